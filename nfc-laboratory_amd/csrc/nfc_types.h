/*
 * nfc_types.h — plain-old-data layouts shared by the host runtime (C-ABI side) and the HIP kernels.
 *
 * Layout philosophy (MI355X): one wavefront = 64 independent capture streams, one lane per stream.
 *   - scalar decoder state: one StreamState record per stream (AoS, touched once per launch),
 *   - history rings: "stream block" storage, [slot][64 lanes] floats, so that a wave whose 64 streams
 *     share the same sample clock touches one contiguous 256-byte row per ring access,
 *   - frames: one packed sink of 32-bit words per context (NfcFrameRecord header + payload), appended
 *     with one atomic per frame; frames of one stream stay in order because one lane emits them.
 *
 * What the state represents follows the reference decoder's data model
 * (src/nfc-lib/lib-lab/lab-radio/src/main/cpp/NfcTech.h:151-393) but only keeps what is ever read:
 * 512-deep sample history instead of 1024 (deepest look-back is NFC-V: 378+94 samples), one shared
 * 256-deep product ring (the reference zeroes integrationData before every listen window), and
 * period-sized correlation rings.
 */
#ifndef NFC_AMD_TYPES_H
#define NFC_AMD_TYPES_H

#include <stdint.h>

#define NFC_LANES 64u          /* streams per stream-block == wavefront width            */
#define NFC_HIST_STORED 512u   /* sample history depth of a stream's rings in HBM (power of two, > 472) */
#ifndef NFC_HIST               /* depth the including translation unit works with: the wave decoder (nfc_wave.hpp) keeps */
#define NFC_HIST NFC_HIST_STORED /* 1024 samples in LDS so that a whole tile can be written ahead of the sample at hand */
#endif
#define NFC_PROD 256u          /* product ring depth for listen-mode integrators (> 189) */
#define NFC_STREAM_BYTES 512u  /* frame assembly buffer, NfcTech.h:288                   */
#define NFC_CORR_MAX 704u      /* bounds NfcConfig::corrTotal for every decodable sample rate (<= ~10.8 MS/s with the stored history depth) */

/* tech / frame enums: values are the reference's wire values (lab-data RawFrame.h:29-84) */
enum
{
   NFC_TECH_NONE = 0,
   NFC_TECH_ANY = 0x0100,
   NFC_TECH_A = 0x0101,
   NFC_TECH_B = 0x0102,
   NFC_TECH_F = 0x0103,
   NFC_TECH_V = 0x0104
};

enum
{
   NFC_FRAME_CARRIER_OFF = 0x0100,
   NFC_FRAME_CARRIER_ON = 0x0101,
   NFC_FRAME_POLL = 0x0102,
   NFC_FRAME_LISTEN = 0x0103
};

enum
{
   NFC_PHASE_CARRIER = 0x0101,
   NFC_PHASE_SELECTION = 0x0102,
   NFC_PHASE_APPLICATION = 0x0103
};

enum
{
   NFC_FLAG_SHORT = 0x01,
   NFC_FLAG_ENCRYPTED = 0x02,
   NFC_FLAG_TRUNCATED = 0x08,
   NFC_FLAG_PARITY = 0x10,
   NFC_FLAG_CRC = 0x20,
   NFC_FLAG_SYNC = 0x40
};

/* symbol timing of one bitrate, in samples (NfcTech.h:168-194) */
struct NfcRate
{
   uint32_t symbolsPerSecond;
   uint32_t p0; /* two symbols  */
   uint32_t p1; /* one symbol   */
   uint32_t p2; /* 1/2 symbol   */
   uint32_t p4; /* 1/4 symbol   */
   uint32_t p8; /* 1/8 symbol   */
   uint32_t delay;    /* symbolDelayDetect */
   uint32_t preamble; /* NFC-F preamble length */
};

/* per-configuration constants, derived on the host exactly like the reference's initialize() chain
 * (NfcDecoder.cpp:295-360, NfcA.cpp:115-212, NfcB.cpp:124-233, NfcF.cpp:108-204, NfcV.cpp:126-234) */
struct NfcConfig
{
   uint32_t sampleRate;
   uint32_t enabled; /* bit0 A, bit1 B, bit2 F, bit3 V */
   double stu;       /* sampleTimeUnit = fs / fc */
   int32_t etu;      /* elementaryTimeUnit */
   float iirA;
   float envW0, envW1;
   float mdevW0, mdevW1;
   float meanW0, meanW1;
   float powerThreshold;
   float lowThreshold, highThreshold;
   float corrThreshold[4]; /* A B F V */
   float minDepth[4];
   float maxDepth[4];
   NfcRate a[3];
   NfcRate b[3];
   NfcRate f[3]; /* index by rate type, [0] unused */
   NfcRate v;
   int32_t vLen2;
   int32_t vLen8;
   int32_t vSlotEnd2[4];
   int32_t vSlotEnd8[256];
   uint32_t corrOffset[6]; /* word offsets of the six correlation rings inside the block ring */
   uint32_t corrTotal;     /* total correlation ring slots */
   uint32_t reserved;
};

/* demodulator state of one (tech, bitrate): the reference's NfcModulationStatus minus its buffers */
struct NfcMod
{
   uint32_t stage;     /* searchModeState    */
   uint32_t winStart;  /* searchStartTime    */
   uint32_t winEnd;    /* searchEndTime      */
   uint32_t sync;      /* searchSyncTime     */
   uint32_t pulses;    /* searchPulseWidth   */
   float thr;          /* searchValueThreshold */
   float phaseThr;     /* searchPhaseThreshold */
   float lastPhase;    /* searchLastPhase    */
   float lastValue;    /* searchLastValue    */
   float syncValue;    /* searchSyncValue    */
   float cD, c0, c1;   /* searchCorrD/0/1Value */
   uint32_t symStart;  /* symbolStartTime    */
   uint32_t symEnd;    /* symbolEndTime      */
   uint32_t symRise;   /* symbolRiseTime     */
   float acc;          /* filterIntegrate    */
   float phaseAcc;     /* phaseIntegrate     */
   float peak;         /* correlatedPeakValue */
   float aux;          /* detectorPeakValue  */
   uint32_t peakTime;  /* correlatedPeakTime */
   uint32_t auxTime;   /* detectorPeakTime   */
};

/* per-technology protocol timing (frameStatus / protocolStatus of the reference); lives in HBM, touched only
 * at frame boundaries */
struct NfcTiming
{
   uint32_t lastCommand;
   uint32_t guardTime;   /* frameStatus.frameGuardTime   */
   uint32_t waitingTime; /* frameStatus.frameWaitingTime */
   uint32_t maxFrameSize;
   uint32_t protoGuardTime;
   uint32_t protoWaitingTime;
};

/* search-mode detector state, one record per (tech, bitrate): only the NfcMod fields a detector reads back
 * between samples. Everything else of the reference's NfcModulationStatus is, while searching, either provably
 * zero (only written in decode mode; the whole record is cleared when the technology resets) or write-only until
 * the lock (searchSyncTime, symbolEndTime, searchPulseWidth ... are then set directly in the working copy).
 * detectorPeakTime of NFC-A/V is never read before it is overwritten and is not tracked. */
struct NfcDetA
{
   uint32_t winStart, winEnd, symStart;
   float acc, peak, aux;
   uint32_t peakTime;
};

struct NfcDetB
{
   uint32_t winStart, winEnd;
   float thr;
   uint32_t symStart, symEnd;
   float aux;
   uint32_t auxTime;
};

struct NfcDetF
{
   uint32_t winStart, winEnd, sync, pulses;
   float thr, lastPhase, lastValue, syncValue, c0;
   uint32_t symStart, symEnd;
   float acc, peak;
   uint32_t peakTime;
};

struct NfcDetV
{
   uint32_t winStart, winEnd, symStart;
   float acc, peak, aux;
   uint32_t peakTime;
};

/* search-mode registers: the eight detector records */
struct NfcSearchRegs
{
   NfcDetA detA[3];
   NfcDetB detB[2];
   NfcDetF detF[2]; /* 212k, 424k */
   NfcDetV detV;
};

/* decode-mode registers: working copy of the locked modulation + symbol / bit stream / frame assembly */
struct NfcDecodeRegs
{
   NfcMod lock;
   NfcRate rt;
   uint32_t lockRate;  /* rate type 0..2 */
   uint32_t pulseCode; /* NFC-V: 0 -> 1 of 4, 1 -> 1 of 256 */
   uint32_t lockBase;  /* correlation ring base of the locked modulation */
   uint32_t lockPos;   /* ring position (idx % rt.p1) of the locked correlator, advanced every sample while locked */
   uint32_t guardEnd;  /* frameStatus.guardEnd / waitingEnd of the locked technology (always written by the poll */
   uint32_t waitingEnd;/* frame's process() before a listen window reads them) */
   uint32_t symPattern, symValue, symStart, symEnd, symEdge;
   uint32_t bsPrevious, bsBits, bsSkip, bsData, bsFlags, bsParity, bsBytes;
   uint32_t frameType, frameRate, frameStart, frameEnd;
   uint32_t maxFrame;  /* protocolStatus.maxFrameSize of the locked technology (register copy of NfcTiming) */
   uint32_t pendType;  /* a frame completed on this sample: NFC_FRAME_POLL / NFC_FRAME_LISTEN, classified and emitted */
   uint32_t pendFlags; /* once, at the end of the decode step (nfc_finish_frame) */
};

/* The part of a stream's state that the kernel keeps in registers for the whole launch. A lane is either searching
 * or decoding, never both, so the two register sets share storage: at lock time the (frozen) detector records are
 * parked in NfcStreamCold::parked and the decode set is initialised in their place; when the technology resets
 * (the only way out of a lock) the detector records are brought back and those of the technology that was locked
 * are cleared, exactly what the reference's resetModulation() does. Decode-set fields have no meaning while
 * searching (the reference zeroes them at reset and rewrites them at the next lock). */
struct NfcStreamState
{
   /* ---- front end (NfcTech.h:317-393) ---- */
   uint32_t clock;       /* signalClock, starts at 0xFFFFFFFF */
   uint32_t pulseFilter;
   float env;            /* signalEnvelope */
   float n1;             /* signalFilterN1 */
   float mdev;           /* signalDeviation */
   float avg;            /* signalAverage */
   float edgePeak;
   uint32_t edgeTime;
   uint32_t carrierOff;
   uint32_t carrierOn;

   uint32_t lockTech;  /* NFC_TECH_* or 0 */
   uint32_t unlock;    /* technology that reset during this decode step (its unparking is done once, at the end of the step) */
   uint32_t bankClock; /* clock of the last sample at which the whole detector bank was stepped (search mode) */
   uint32_t chainedA;  /* NFC-A chained frame flags (Encrypted after AUTH) */
   uint32_t served;    /* NfcLaunch::launchSeq of the last launch that advanced this stream: a block is taken by exactly
                          one of the common / exact-modulo kernels of a launch, whichever sees it first */

   /* ---- ring positions (idx % period) of every correlator ---- */
   uint32_t posA[3];
   uint32_t posF[2]; /* 212k, 424k */
   uint32_t posV1;   /* mod p1 */
   uint32_t posV0;   /* mod p0 */

   union
   {
      NfcSearchRegs search;
      NfcDecodeRegs decode;
   } u;
};

/* What a lane of the time-parallel path found out about the pulse memory of an NFC-F preamble detector (pulse counter and
 * threshold of the last pulse: NfcCarry::pulsesF / thrF) while it ran on the values it had ASSUMED: every evaluation of a
 * pulse (nfcf_track_preamble) before the record starts over narrows the set of values the lane could have been given
 * without deciding anything differently. All zero = no evaluation yet, nothing required. */
struct NfcFBound
{
   uint32_t lowMax;   /* 1 + the largest counter value an evaluation found below 94 (0: none): the true one must be below 94 there too */
   uint32_t highMin;  /* 1 + the smallest counter value an evaluation found at 94 or above (0: none): ... and not below 94 there */
   float thrAbove;    /* NFC_FBOUND_ABOVE: the true threshold must be greater than this (the pulse was below the threshold: record cleared) */
   float thrBelow;    /* NFC_FBOUND_BELOW: ... smaller than this (the pulse was above it: accepted) */
   uint32_t flags;
};

#define NFC_FBOUND_ABOVE 1u
#define NFC_FBOUND_BELOW 2u
#define NFC_FBOUND_EXACT 4u   /* an evaluation the bounds do not describe (the pulse equal to the threshold, a completed preamble): only the assumed values will do */
#define NFC_FBOUND_THR_OWN 8u /* the threshold has been set or cleared by the lane since it started: no longer the assumed one */

/* the part that stays in HBM and is only touched at frame boundaries */
struct NfcStreamCold
{
   NfcTiming tim[4];     /* A B F V */
   NfcSearchRegs parked; /* detector records while a technology is locked */
   NfcDecodeRegs init;   /* decode register set prepared by the detector that locked, installed by nfc_enter_lock */
   uint32_t framesOut;
   uint32_t lastUnlock; /* clock of the last return to search mode */
   uint32_t bankRun;    /* clock at which the detector bank began its current unbroken run of steps (no lock, no sample
                           below the power threshold since): a lane of the time-parallel path may only stop once the
                           correlation rings hold nothing older than that */
   uint32_t emitClock;  /* clock of the last carrier frame (the decoder zeroes edgeTime when it emits one) */
   uint32_t emitValid;
   uint32_t emitOwn;    /* a lane's note (time-parallel path): it has emitted a carrier frame itself. One that has not never looked at the
                           record of the last carrier frame it was given - that record only zeroes the edge time a carrier frame is
                           stamped with - and hands on the one the stream really holds (nfc_chain_follow) */
   uint32_t trackedEnd; /* a lane's note: the edge tracker's time where the lane ended (what its edge time is formed from: nfc_edge_time) */
   uint32_t frameHead;  /* chained frame records of this lane in the staging sink: word offset + 1 of the first / last */
   uint32_t frameTail;
   /* Ring phase labels of the seven correlation ring positions (A106 A212 A424 F212 F424 V(p1) V(p0)):
    * position = (reference position + label) % period. Zero for a stream decoded sequentially from its start. A lane of
    * the time-parallel path numbers its rings from zero at its first sample so that the lanes of a wave, whatever their
    * clocks, touch the same ring rows; the state it leaves keeps that numbering. Only the exact-modulo kernel variants
    * (stream start, 32-bit clock wrap) compute positions from the clock and have to add it. */
   uint32_t label[7];
   uint32_t clearedF[2]; /* an NFC-F preamble detector cleared its pulse counter since the lane started (NfcCarry::pulsesF) */
   NfcFBound boundF[2];  /* ... and what the evaluations before that require of the pulse memory the lane was given */
   /* The waiting time of a technology's protocol (NfcTiming::protoWaitingTime: set by an ATS / ATQB, put back by a REQA / HLTA /
    * REQB / REQC) reaches a lane's decode through one comparison only: `clock > waitingEnd` while the start of an answer is
    * looked for (nfc*_listen_*start), waitingEnd = end of the poll frame + waiting time. A lane of the time-parallel path that
    * inherited the value notes what its waits on it required (nfc_wait_*, nfc_core.hpp), so that the chain can tell whether
    * another value would have made any of those comparisons come out differently (nfc_chain_follow):
    *   waitUsed[t]  the longest such wait that ended by something else than the time running out - an answer, a modulation too
    *                deep - counted from the sample the waiting time counts from: any waiting time of at least that much leaves
    *                every comparison of those waits false, as it was;
    *   waitFlags    bit t: the wait at hand of technology t runs on the inherited value; bit 4 + t: the lane has set the
    *                technology's protoWaitingTime itself (what it was given no longer matters from there on); bit 8 + t: a wait
    *                on the inherited value ran out (only the very same value reproduces that) */
   uint32_t waitUsed[4];
   uint32_t waitFlags;
   uint32_t usedTech;    /* bit t: technology t (A B F V) has been locked since the lane started. The protocol timing of a
                            technology is only read when it locks and while its frames are processed, so a lane that never
                            locked it neither depends on what it assumed there nor changes it (nfc_chain_follow).
                            bit 4+t: a listen frame of t was classified by a lastCommand the lane had not written itself
                            (poll frames are classified by their first byte and write it: nfc*_process); bit 8+t: the
                            lane has set lastCommand of t; bit 18+t: a frame of t has been processed; bit 22+t: the first
                            one was the command that starts the technology's protocol over (nfc_finish_frame); bit 12+i: NFC-F preamble detector i evaluated a pulse with the
                            counter / threshold it had inherited (nfcf_track_preamble); bit 14+i: its tracker ran on the
                            record the lane had inherited; bit 16+i: the lane has (partially) reset that record */
};

/* header of one frame in the frame sink, followed by (length+3)/4 payload words */
struct NfcFrameRecord
{
   uint32_t stream;
   uint32_t tech;
   uint32_t type;
   uint32_t flags;
   uint32_t phase;
   uint32_t rate;
   uint32_t start;
   uint32_t end;
   uint32_t length;
};

#define NFC_FRAME_HEADER_WORDS 9u
#define NFC_FRAME_MAX_WORDS (NFC_FRAME_HEADER_WORDS + NFC_STREAM_BYTES / 4u)

#endif
