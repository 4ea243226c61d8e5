/*
 * nfc_scan.h — records of the time-parallel path (scan kernel -> window builder -> windowed decode -> chain),
 * shared by the host runtime and the kernels.
 *
 * Why it exists. The reference decoder (NfcDecoder.cpp:394-418) walks every sample through the front end and, while no
 * technology is locked, through eight detectors. Two facts make most of that walk redundant on quiet signal:
 *
 *  - the front end (NfcTech.cpp:28-105: envelope, DC filter, mean deviation, average, edge-peak tracker) depends on the
 *    samples only, never on detector or lock state, and its recurrences are contractive: started a few thousand samples
 *    early from a guess they reach the very bit pattern of the true state. A capture can therefore be cut into chunks
 *    that are walked independently, each seam checked bit for bit against the true end state of the chunk before it;
 *  - a detector of the search bank at rest (cleared record) can only leave that state when its correlation exceeds a
 *    fraction of the envelope (NfcA.cpp:236-300, NfcF.cpp:247-300, NfcV.cpp:262-300), and that correlation is bounded
 *    by the spread of the raw signal over its window: |S0 - S1| <= 2 * W * (max - min). Where the spread of the last
 *    568 samples stays below 0.49 * threshold * envelope no detector can move; NFC-B (NfcB.cpp:262-300) needs a falling
 *    edge of the DC-removed signal below -depth * envelope. Both are decided per tile of 64 samples during the walk.
 *
 * The decoder proper (the per-sample step machine of nfc_core.hpp) is then only run over "windows": from a tile where
 * something may happen until the machine is back at rest, every window on its own lane, started from the scanned
 * front-end state plus a short warm-up that refills the history rings. Whatever a window assumed about the state the
 * previous one left behind (protocol timing, carrier state) is verified afterwards by the chain kernel and the window
 * is run again if the assumption was wrong. Streams whose samples are not on the int16 grid of the WAV captures
 * (arbitrary fp32 magnitudes: there the reference's running sums carry rounding that depends on their whole history),
 * whose seams do not verify, or that come near the 32-bit clock wrap take the sequential path as before.
 */
#ifndef NFC_AMD_SCAN_H
#define NFC_AMD_SCAN_H

#include <stdint.h>

#include "nfc_types.h"

#define NFC_SCAN_TILE 64u     /* samples per tile flag word */
#define NFC_SCAN_POINT 512u   /* samples between stored front-end states (== history ring depth: window starts align) */
#define NFC_SCAN_LOOKBACK 9u  /* previous tiles whose spread can still reach a detector: NFC-V looks 378+94+95 back */
#define NFC_SCAN_EDGEBACK 2u  /* previous tiles an NFC-B edge can come from (212k detector: one 106k symbol back) */

/* tile flag bits (low byte: reasons to run the decoder over the tile) */
#define NFC_TILE_RANGE 0x01u    /* spread of the raw signal large enough for an NFC-A/F/V correlator to exceed its threshold */
#define NFC_TILE_EDGE 0x02u     /* falling edge of the DC-removed signal beyond the NFC-B threshold */
#define NFC_TILE_UNARMED 0x04u  /* envelope below the power threshold (detectors not stepped) or decoder younger than 1024 samples */
#define NFC_TILE_CARRIER 0x08u  /* the average crosses into the other carrier zone: a carrier frame may be due */
#define NFC_TILE_OFFGRID 0x10u  /* a sample that is not a multiple of 2^-15 (or beyond +-4): box sums not order-independent */
#define NFC_TILE_NOHISTORY 0x20u /* look-back reaches before the first sample of this submission */
#define NFC_TILE_REWALKED 0x40u /* the scanned envelope was wrong here (seam did not verify) and has been walked again */
#define NFC_TILE_BUSY 0x3Fu
#define NFC_TILE_RETIRE_OK 0x100u /* set by the window builder: no busy tile for at least NFC_WINDOW_GAP tiles from here */
#define NFC_TILE_DARK 0x200u      /* every sample below the power threshold and no carrier event: the decoder only runs its front end */
#define NFC_TILE_DARK_RUN_SHIFT 16 /* bits 16..31: number of dark tiles in a row from this one on (saturating) */
#define NFC_DARK_JUMP 24u         /* dark tiles ahead that make a searching lane jump (it lands NFC_SCAN_POINT samples before they end) */

#define NFC_WINDOW_GAP 16u        /* quiet tiles that separate two windows (1024 samples) */
#define NFC_WINDOW_WARM_FRONT 512u /* samples of front end only at the start of a window lane: refills the sample history */
#define NFC_WINDOW_WARM_CORR 256u  /* then samples of correlator upkeep only: refills every search correlation ring (p1 <= 189) */
#define NFC_WINDOW_SETTLE 256u     /* unbroken detector-bank steps before a lane may retire at rest (stale ring entries) */
#define NFC_WINDOW_CUT 16384u      /* inside busy signal a new lane starts this often ... */
#define NFC_WINDOW_VERIFY 2048u    /* ... and this long after going live it publishes its state: a lane before it whose
                                      state is then the same hands over to it (both need NFC_WINDOW_STEADY steady steps) */
#define NFC_WINDOW_STEADY 1024u    /* unbroken detector-bank steps that make the correlation rings a function of the samples alone */

/* front-end state before sample `NFC_SCAN_POINT * k` of a submission (after the sample before it) */
struct NfcScanPoint
{
   float env, n1, mdev, avg, edgePeak;
   uint32_t pulseFilter;
   uint32_t edgeTime; /* clock of the last update of the edge-peak tracker (the decoder zeroes its copy when it emits a carrier frame) */
   uint32_t zone;     /* carrier zone the average was last seen in: 1 above the high threshold, 2 below the low one, 0 neither yet */
};

/* what the walk records per tile of 64 samples; the tile tests (with their look-back) are evaluated per stream afterwards,
 * once the envelope is known to be the true one everywhere */
struct NfcScanTile
{
   float xmin, xmax; /* raw signal */
   float fmin;       /* DC-removed signal */
   float envmin;     /* envelope */
   float envmax;
   uint32_t bits;    /* NFC_TILE_UNARMED (young decoder) | NFC_TILE_CARRIER | NFC_TILE_OFFGRID seen by the walk */
};

/* what a chunk walker started from (after its warm-up) and ended with: seam k is sound when start(k) == end(k-1) */
struct NfcScanSeam
{
   NfcScanPoint start;
   NfcScanPoint end;
};

/* one stream of a submission */
struct NfcScanJob
{
   const uint8_t *data; /* device pointer, count*stride floats */
   uint32_t count;      /* samples */
   uint32_t slot;       /* stream slot holding the state the submission starts from */
   uint32_t firstChunk; /* index of its first chunk in the chunk table */
   uint32_t chunks;
   uint32_t firstTile;  /* index of its first tile flag word */
   uint32_t firstPoint; /* index of its first NfcScanPoint (points 0 .. count / NFC_SCAN_POINT) */
   uint32_t firstWindow; /* filled by the window builder: windows of this job are contiguous, ordered by activation */
   uint32_t windows;
   uint32_t status;     /* NFC_JOB_* bits, written by the kernels */
   uint32_t finalLane;  /* virtual slot whose state is the stream's state after the submission (chain kernel) */
   uint32_t passes;
   uint32_t busyTiles;  /* tiles kernel: tiles with something for the decoder to do */
   uint32_t cut;        /* inside busy signal a new lane starts this often (NFC_WINDOW_CUT, or more when the submission has lanes to spare) */
};

#define NFC_JOB_OFFGRID 0x01u   /* samples off the int16 grid: sequential path */
#define NFC_JOB_SEAM 0x02u      /* a chunk seam did not verify: sequential path */
#define NFC_JOB_RERUN 0x04u     /* the chain kernel asked for another pass */
#define NFC_JOB_GIVEUP 0x08u    /* too many passes: sequential path */
#define NFC_JOB_OVERFLOW 0x10u  /* window table full: sequential path */
#define NFC_JOB_ALONE 0x40u     /* the stream's carry lane decodes it alone (no windows, one pass): samples off the capture grid */
#define NFC_JOB_OFFGRID_SEEN 0x80u /* samples off the int16 grid in a stream that stays on the path (NfcScanParams::offGridAlone): no windows */
#define NFC_JOB_INVALID (NFC_JOB_OFFGRID | NFC_JOB_SEAM | NFC_JOB_GIVEUP | NFC_JOB_OVERFLOW)

struct NfcScanChunk
{
   uint32_t job;
   uint32_t index; /* chunk number inside the job; bit 31: walk it again from the true end state of the chunk before (no warm-up) */
};

#define NFC_ZONE_FOLLOWS 0x400u /* bit of NfcScanSeam::start.zone (beside the carrier zone and the edge-time bits of nfc_scan.hpp): the chunk is listed for
                                   a second walk of its envelope tracker alone, like the chunk before it; nfc_envelope_kernel walks it in the
                                   same go as that one (nfc_seams_check, nfc_envelope.hpp) */
#define NFC_ZONE_LISTED 0x800u  /* ... the chunk is on one of this round's lists at all (whoever walks it owns its records this round) */
#define NFC_CHUNK_REPAIR 0x80000000u
#define NFC_CHUNK_ENVELOPE 0x40000000u /* with NFC_CHUNK_REPAIR: only the envelope tracker (envelope, pulse counter) started wrong; the other
                                          recurrences do not depend on it and stand as walked: the second walk is the tracker's alone */

/* state a window inherits from whatever ran before it on the stream, apart from the front end (scanned) and the
 * history / correlation rings (rebuilt by the warm-up): compared field by field by the chain kernel */
struct NfcCarry
{
   NfcTiming tim[4];
   uint32_t chainedA;
   uint32_t carrierOn;
   uint32_t carrierOff;
   uint32_t emitClock; /* clock of the last carrier frame the decoder emitted (it zeroes edgeTime then) */
   uint32_t emitValid;
   uint32_t edgeTime;  /* the decoder's edge time (compared when two lanes meet at the same sample) */
   /* what an NFC-F preamble detector at rest still remembers: its partial resets (modulation deeper than the NFC-F
    * maximum, peak timeout: NfcF.cpp:262-283) leave the pulse counter and the threshold of the last pulse in place, and
    * every 100 % ASK pause of an NFC-A / NFC-V frame goes through one; both are read again when the next pulse ends */
   uint32_t pulsesF[2];
   float thrF[2];
   uint32_t clearedF[2]; /* result only: the lane's detector started its pulse count over (not compared) */
   uint32_t ownF[2];     /* result only: ... or set the threshold of the last pulse itself (NFC_FBOUND_THR_OWN) */
   uint32_t emitOwn;     /* result only: the lane has emitted a carrier frame itself (NfcStreamCold::emitOwn) */
   uint32_t waitOwn;     /* result only: NfcStreamCold::waitFlags as the lane held them (bit 4 + t: it has set protoWaitingTime of t itself) */
   /* The detector records (running sums apart). A lane starts with all of them at rest; one that was tracking something
    * when another technology locked comes back from the lock with its window in the past and stays like that until the
    * next strong pulse (NfcF.cpp:262-283 and the like): a state no amount of warm-up reproduces, so it travels here. */
   NfcSearchRegs search;
};

/* one window: a lane of the windowed decode launch */
struct NfcWindow
{
   uint32_t job;
   uint32_t start;    /* first sample the lane consumes (multiple of NFC_SCAN_POINT; 0 for the lane that carries the stream's state in) */
   uint32_t activate; /* sample at which the lane's decoder goes live (multiple of NFC_SCAN_TILE) */
   uint32_t verify;   /* sample at which the lane publishes its state (activate + NFC_WINDOW_VERIFY), 0xFFFFFFFF: never */
   uint32_t stop;     /* out: first sample the lane did not consume */
   uint32_t retired;  /* out: 0 ran to the end of the submission, 1 stopped at rest, 2 handed over at `stop` to lane handTo */
   uint32_t handTo;
   uint32_t noHand;   /* chain kernel: when run again, do not hand over to lanes up to this index */
   uint32_t rerun;    /* chain kernel: run this window again (with `want`) */
   uint32_t live;     /* chain kernel: its frames (after liveFrom) are the stream's frames */
   uint32_t liveFrom; /* chain kernel: 0 all its frames, else frameTail marker: only the records chained after that one */
   uint32_t pubState; /* 0 nothing published (yet), 1 state published, 2 reached `verify` in a state that cannot be compared */
   uint32_t pubTail;  /* the lane's frameTail when it published */
   uint32_t pubDigest[2];
   uint32_t stopDigest[2]; /* digest of the lane's own state where it handed over */
   uint32_t saved;    /* out: 1 + slot of the save area holding the lane's rings as it left them (lanes that ran to the end of the submission), 0: none */
   uint32_t tracked;  /* edge-tracker time at the lane's first sample (scanned): with the carry's last carrier frame it gives the decoder's edge time */
   int32_t pulsesFix[2]; /* chain kernel: the lane ran on an NFC-F pulse counter that was off by this much, without consequence (NfcFBound): what it leaves is put right by it */
   uint32_t thrPass[2];  /* ... and never set the threshold of the last pulse: it leaves the one the stream held */
   NfcCarry carry;    /* what the lane assumed when it last ran */
   NfcCarry want;     /* chain kernel: what it has to assume in the next pass (rerun) */
   NfcCarry pubCarry; /* the lane's carry when it published */
};

#endif
