/*
 * nfc_core.hpp — the demodulator as a per-sample step machine, one lane per capture stream.
 *
 * This header is device code: kernels.hip includes it with NFC_DEV = `__device__ __forceinline__`.
 * (tests/hostsim compiles the same text with NFC_DEV = `static inline` to unit-test the state
 * machine on a CPU-only box; that build is test infrastructure and is never linked into the
 * product library.)
 *
 * Behavioural contract: identical frames to the reference CPU decoder
 *   lab::NfcDecoder::nextFrames            src/nfc-lib/lib-lab/lab-radio/src/main/cpp/NfcDecoder.cpp:374-467
 *   NfcDecoderStatus::nextSample           .../cpp/NfcTech.cpp:28-105
 *   NfcA / NfcB / NfcF / NfcV              .../cpp/tech/Nfc{A,B,F,V}.cpp
 * The reference runs nested loops (frame -> symbol -> sample) that return when the buffer is
 * exhausted and recompute their indices from signalClock on re-entry; every sample is therefore
 * handled by exactly one "mode" after the front end. Here that is made explicit: nfc_step()
 * advances one sample: front end, then either the search bank (all enabled detectors, first lock
 * wins) or the locked technology's symbol machine, whose symbols feed bit/byte/frame assembly.
 * All arithmetic is fp32 without contraction (compile with -ffp-contract=off) and unsigned 32-bit
 * sample clocks, matching the reference's Release build (-msse3 -mno-avx: no FMA).
 */
#ifndef NFC_AMD_CORE_HPP
#define NFC_AMD_CORE_HPP

#include "nfc_types.h"

#ifndef NFC_DEV
#error "define NFC_DEV before including nfc_core.hpp"
#endif
#ifndef NFC_ATOMIC_ADD
#error "define NFC_ATOMIC_ADD(ptr, value) (returns the old value) before including nfc_core.hpp"
#endif

/* Divergent control flow is written as a sequence of independent `if` regions, never as if / else-if chains over
 * regions that update the stream state: the AMDGPU back end linearises `if (a) X else Y` into X then Y, the values
 * X produced and the originals Y still reads are live together, and every state register one arm leaves untouched
 * is copied (v_mov) on the other, at every nesting level. Mutually exclusive predicates are taken from a snapshot
 * and passed through NFC_OPAQUE so that the optimiser cannot fold the sequence back into a chain. */
#ifdef __HIP_DEVICE_COMPILE__
#define NFC_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define NFC_OPAQUE(x) asm volatile("" : "+r"(x))
#endif

/* true when the predicate holds for any stream of the block (wave-uniform on the device) */
#ifndef NFC_ANY
#error "define NFC_ANY(predicate) before including nfc_core.hpp"
#endif

/* Marks a freshly loaded value as used at this point, so that the wait for the load is placed here (inside the rare
 * branch that issued it) and not where the branch rejoins the common path. */
#ifdef __HIP_DEVICE_COMPILE__
#define NFC_ARRIVED(x) asm volatile("" : "+v"(x))
#else
#define NFC_ARRIVED(x) ((void)(x))
#endif

/* Wait for every outstanding memory operation. Used after the bulk state reloads of the rare lock / unlock events and
 * after the state record is read at kernel start: left outstanding into the sample loop, each of those ~100 registers
 * would carry a conservative wait (covering stores as well, see NFC_ARRIVED) to wherever it is first touched in the
 * next step - on the common path. */
#ifdef __HIP_DEVICE_COMPILE__
#define NFC_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70) /* vmcnt(0) */
#else
#define NFC_DRAIN() ((void)0)
#endif

/* Where the rings live and how they are laid out is the includer's choice: the stream-parallel kernels keep them in
 * HBM as [slot][64 lanes] (one lane per stream), the wave decoder (nfc_wave.hpp: one wave per stream) keeps one stream's
 * rings in LDS, [slot] only. */
#ifndef NFC_RING_FLOAT
#define NFC_RING_FLOAT float
#endif
#ifndef NFC_RING_STRIDE
#define NFC_RING_STRIDE NFC_LANES
#endif

/* per-lane view of the stream-block storage; every ring pointer is already offset by the lane */
struct NfcLaneMem
{
   NFC_RING_FLOAT *ring; /* stream-block ring storage (wave-uniform base), regions below, each [slots][NFC_RING_STRIDE] */
   uint32_t lane;  /* this stream's column */
   bool exact;     /* take ring positions by exact modulo (stream start / 32-bit clock wrap) instead of incrementally */
   uint8_t *bytes; /* frame assembly buffer, NFC_STREAM_BYTES contiguous */
   uint32_t *sink;       /* frame sink shared by every stream of the launch (packed records) */
   uint32_t *sinkCursor; /* words used, advanced atomically */
   uint32_t *sinkDropped;
   uint32_t sinkWords;
   uint32_t streamId;
   NfcStreamCold *cold;  /* protocol timing of this stream (HBM) */
   const NfcConfig *tables; /* configuration in memory, for its dynamically indexed tables (NFC-V pulse slots) */
   bool linked;          /* frame records are chained per lane in a staging sink (time-parallel path) */
   uint32_t *flags;      /* time-parallel lanes only (linked): what the lane has looked at of the state it inherited, bits of
                            NfcStreamCold::usedTech; kept in fast storage during the run (the sequential kernels track nothing) */
#ifdef NFC_F_DEEP_CTX
   NFC_F_DEEP_CTX deep;  /* where NFC_F_DEEP finds what the shortened histories no longer hold */
#endif
};

/* The histories of the filtered signal, its mean deviation and the modulation depth may be kept shorter than the raw
 * samples' (NFC_HIST_F < NFC_HIST: the wave decoder, whose rings are LDS): everything but NFC-V looks back less than 192
 * samples there (NFC-A 424k BPSK: delay 141 + one symbol of 24; NFC-A / -B depth taps: 153), NFC-V looks back up to 402.
 * An includer that shortens them provides NFC_F_DEEP(mem, region, sampleClock) for the reads that may reach further
 * (from the front end's planes in HBM: nfc_wave.hpp). */
#ifndef NFC_HIST_F
#define NFC_HIST_F NFC_HIST
#endif

/* ring regions (in slots) inside a stream block */
#define NFC_R_X 0u                                      /* samplingValue  [NFC_HIST]   */
#define NFC_R_FILT (NFC_HIST)                           /* filteredValue  [NFC_HIST_F] */
#define NFC_R_MDEV (NFC_HIST + NFC_HIST_F)              /* meanDeviation  [NFC_HIST_F] */
#define NFC_R_DEPTH (NFC_HIST + 2u * NFC_HIST_F)        /* modulateDepth  [NFC_HIST_F] */
#define NFC_R_PROD (NFC_HIST + 3u * NFC_HIST_F)         /* listen-mode product ring [NFC_PROD] */
#define NFC_R_CORR (NFC_HIST + 3u * NFC_HIST_F + NFC_PROD) /* correlation rings [corrTotal] */

/* 32-bit index from a wave-uniform base: the access becomes `global_load v, v_off, s[base]` */
#define NFC_AT(m, region, slot) ((m).ring[((region) + (uint32_t)(slot)) * NFC_RING_STRIDE + (m).lane])
#define NFC_HMASK (NFC_HIST - 1u)
#define NFC_FMASK (NFC_HIST_F - 1u)

/* a read of the filtered / deviation / depth history at the sample of clock `clk`, which may lie further back than the
 * shortened history holds (NFC-V); `wanted` false: the value is not used (the stream-parallel kernels then read a
 * harmless row every lane reads anyway, `common`, instead of branching) */
#ifdef NFC_F_DEEP
#define NFC_F_AT(mem, region, clk) NFC_F_DEEP((mem), (region), (clk))
#define NFC_F_TAP(mem, wanted, region, clk, common) ((wanted) ? NFC_F_DEEP((mem), (region), (clk)) : 0.0f)
#else
#define NFC_F_AT(mem, region, clk) NFC_AT((mem), (region), (clk) & NFC_FMASK)
#define NFC_F_TAP(mem, wanted, region, clk, common) NFC_AT((mem), 0u, (wanted) ? (region) + ((clk) & NFC_FMASK) : (common))
#endif

/* the raw sample of clock `sampleClock`, for the deepest look-back of the path (NFC-V: 472 samples): the wave decoder,
 * which writes a whole tile ahead into a history exactly NFC_HIST_STORED deep, keeps the samples that tile displaced */
#ifndef NFC_X_OLD_INDEX
#define NFC_X_OLD_INDEX(mem, sampleClock) (NFC_R_X + ((sampleClock) & NFC_HMASK)) /* slot index from the start of the ring storage */
#endif
#define NFC_PMASK (NFC_PROD - 1u)

/* symbol patterns (private numbering; 0 = nothing yet, 1 = give up / timeout) */
enum
{
   SYM_NONE = 0,
   SYM_TIMEOUT = 1,
   /* NFC-A */
   A_X = 2, A_Y = 3, A_Z = 4, A_D = 5, A_E = 6, A_F = 7, A_M = 8, A_N = 9, A_S = 10, A_O = 11,
   /* NFC-B */
   B_L = 2, B_H = 3, B_S = 4, B_M = 5, B_N = 6, B_O = 7,
   /* NFC-F */
   F_L = 2, F_H = 3, F_S = 4, F_E = 5,
   /* NFC-V */
   V_0 = 2, V_1 = 3, V_2 = 4, V_8 = 5, V_S = 6, V_E = 7
};

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

NFC_DEV float nfc_abs(float v)
{
   return __builtin_fabsf(v);
}

NFC_DEV uint32_t nfc_bits(float v)
{
   return __builtin_bit_cast(uint32_t, v);
}

/* samples for a duration given in 1/fc units: static_cast<int>(sampleTimeUnit * units) */
NFC_DEV uint32_t nfc_tu(const NfcConfig &c, int units)
{
   return (uint32_t)(int)(c.stu * (double)units);
}

NFC_DEV void nfc_mod_clear(NfcDetA &m)
{
   m.winStart = 0; m.winEnd = 0; m.symStart = 0;
   m.acc = 0; m.peak = 0; m.aux = 0; m.peakTime = 0;
}

NFC_DEV void nfc_mod_clear(NfcDetB &m)
{
   m.winStart = 0; m.winEnd = 0; m.thr = 0;
   m.symStart = 0; m.symEnd = 0; m.aux = 0; m.auxTime = 0;
}

NFC_DEV void nfc_mod_clear(NfcDetF &m)
{
   m.winStart = 0; m.winEnd = 0; m.sync = 0; m.pulses = 0;
   m.thr = 0; m.lastPhase = 0; m.lastValue = 0; m.syncValue = 0; m.c0 = 0;
   m.symStart = 0; m.symEnd = 0; m.acc = 0; m.peak = 0; m.peakTime = 0;
}

NFC_DEV void nfc_mod_clear(NfcDetV &m)
{
   m.winStart = 0; m.winEnd = 0; m.symStart = 0;
   m.acc = 0; m.peak = 0; m.aux = 0; m.peakTime = 0;
}

NFC_DEV void nfc_mod_clear(NfcMod &m)
{
   m.stage = 0; m.winStart = 0; m.winEnd = 0; m.sync = 0; m.pulses = 0;
   m.thr = 0; m.phaseThr = 0; m.lastPhase = 0; m.lastValue = 0; m.syncValue = 0;
   m.cD = 0; m.c0 = 0; m.c1 = 0;
   m.symStart = 0; m.symEnd = 0; m.symRise = 0;
   m.acc = 0; m.phaseAcc = 0; m.peak = 0; m.aux = 0;
   m.peakTime = 0; m.auxTime = 0;
}

NFC_DEV void nfc_zero_ring(const NfcLaneMem &mem, uint32_t from, uint32_t count)
{
#pragma clang loop unroll(disable)
   for (uint32_t i = 0; i < count; i++)
      NFC_AT(mem, 0u, from + i) = 0.0f;
}

/* what the reference does to the locked modulation at the end of every poll frame
 * ("clear modulation status for receiving card response", e.g. NfcA.cpp:491-511) */
NFC_DEV void nfc_poll_end_clear(const NfcLaneMem &mem, NfcMod &m, uint32_t corrFrom, uint32_t corrCount)
{
   m.symStart = 0; m.symEnd = 0; m.acc = 0; m.phaseAcc = 0; m.stage = 0;
   m.sync = 0; m.winStart = 0; m.winEnd = 0; m.pulses = 0;
   m.lastValue = 0; m.lastPhase = 0; m.thr = 0; m.phaseThr = 0; m.peak = 0;
   nfc_zero_ring(mem, NFC_R_PROD, NFC_PROD);
   nfc_zero_ring(mem, NFC_R_CORR + corrFrom, corrCount);
}

NFC_DEV void nfc_clear_assembly(NfcStreamState &s)
{
   s.u.decode.bsPrevious = 0; s.u.decode.bsBits = 0; s.u.decode.bsSkip = 0;
   s.u.decode.bsData = 0; s.u.decode.bsFlags = 0; s.u.decode.bsParity = 0; s.u.decode.bsBytes = 0;
}

NFC_DEV void nfc_push_byte(const NfcLaneMem &mem, NfcStreamState &s, uint32_t value)
{
   /* the reference's buffer is 512 bytes (NfcTech.h:288); beyond that it would overrun, we drop */
   if (s.u.decode.bsBytes < NFC_STREAM_BYTES)
      mem.bytes[s.u.decode.bsBytes] = (uint8_t)value;
   s.u.decode.bsBytes++;
}

NFC_DEV uint32_t nfc_byte(const uint8_t *data, uint32_t len, uint32_t i)
{
   return i < len ? data[i] : 0u;
}

/* CRC-16/CCITT as lab::Crc::ccitt16 (lab-data Crc.cpp:96-113), computed bitwise */
NFC_DEV uint32_t nfc_crc16(const uint8_t *data, uint32_t count, uint32_t init, bool reflected)
{
   if (count == 0)
      return (~init) & 0xFFFFu;

   uint32_t crc = init & 0xFFFFu;

   if (reflected)
   {
#pragma clang loop unroll(disable)
      for (uint32_t i = 0; i < count; i++)
      {
         crc ^= data[i];
#pragma clang loop unroll(disable)
         for (int k = 0; k < 8; k++)
            crc = (crc & 1u) ? (crc >> 1) ^ 0x8408u : (crc >> 1);
      }
   }
   else
   {
#pragma clang loop unroll(disable)
      for (uint32_t i = 0; i < count; i++)
      {
         crc ^= ((uint32_t)data[i]) << 8;
#pragma clang loop unroll(disable)
         for (int k = 0; k < 8; k++)
            crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x1021u) & 0xFFFFu : (crc << 1) & 0xFFFFu;
      }
   }

   return crc;
}

/* append one frame to the launch's frame sink: [stream, tech, type, flags, phase, rate, start, end, length, payload...] */
NFC_DEV void nfc_emit(const NfcLaneMem &mem, NfcStreamState &s, uint32_t tech, uint32_t type, uint32_t flags,
                      uint32_t phase, uint32_t rate, uint32_t start, uint32_t end, const uint8_t *data, uint32_t len)
{
   if (len > NFC_STREAM_BYTES)
      len = NFC_STREAM_BYTES;

   /* a lane of the time-parallel path chains its records: one link word in front of each (see nfc_finish_frames) */
   const uint32_t link = mem.linked ? 1u : 0u;
   const uint32_t words = NFC_FRAME_HEADER_WORDS + ((len + 3u) >> 2) + link;
   uint32_t at = NFC_ATOMIC_ADD(mem.sinkCursor, words);

   mem.cold->framesOut++;

   /* a record is only written when a maximum-size record would still fit: everything that starts at or
    * below sinkWords - NFC_FRAME_MAX_WORDS is valid, everything above was dropped (no holes to guess) */
   if (at + NFC_FRAME_MAX_WORDS > mem.sinkWords)
   {
      NFC_ATOMIC_ADD(mem.sinkDropped, 1u);
      return;
   }

   if (link)
   {
      mem.sink[at] = 0;
      if (mem.cold->frameTail)
         mem.sink[mem.cold->frameTail - 1u] = at + 1u;
      else
         mem.cold->frameHead = at + 1u;
      mem.cold->frameTail = at + 1u;
      at++;
   }

   uint32_t *w = mem.sink + at;

   w[0] = mem.streamId;
   w[1] = tech; w[2] = type; w[3] = flags; w[4] = phase;
   w[5] = rate; w[6] = start; w[7] = end; w[8] = len;

#pragma clang loop unroll(disable)
   for (uint32_t i = 0; i < len; i += 4)
   {
      uint32_t v = 0;
#pragma clang loop unroll(disable)
      for (uint32_t k = 0; k < 4 && i + k < len; k++)
         v |= ((uint32_t)data[i + k]) << (8 * k);
      w[NFC_FRAME_HEADER_WORDS + (i >> 2)] = v;
   }
}

/* ------------------------------------------------------------------------------------------ */
/* front end: NfcDecoderStatus::nextSample, NfcTech.cpp:28-105                                */
/* ------------------------------------------------------------------------------------------ */

/* what the front end derives from the current sample (the reference reads these back from its ring) */
struct NfcNow
{
   float x;     /* samplingValue  */
   float filt;  /* filteredValue  */
   float mdev;  /* meanDeviation  */
   float depth; /* modulateDepth  */
};

/* The envelope tracker on its own (NfcTech.cpp:36-56): follows the signal while it stays within 5 % of the envelope,
 * otherwise only once per ten symbols. Unlike the other recurrences of the front end it is not contractive (which
 * samples update it depends on its own value), so the scan path cannot always reach its true state from a guess and
 * has to be able to walk it alone (nfc_scan.hpp: nfc_envelope_fix). `pulseFilter` has already been incremented. */
NFC_DEV void nfc_envelope_step(const NfcConfig &c, uint32_t clock, uint32_t &pulseFilter, float &envelope, float value)
{
   float env = envelope;

   /* reference: |x - env| / env < 0.05f. Decided without the division unless the ratio is within 0.2 % of the
    * limit (for env > 0: dev < 0.0499*env implies fl(dev/env) < 0.05f, dev > 0.0501*env implies the opposite).
    * Written with selects: the only branch left is the rare division. */
   const float dev = nfc_abs(value - env);
   const bool below = dev < 0.0499f * env;
   const bool above = dev > 0.0501f * env;
   bool tracking = below;

   if (!(env > 0.0f && (below || above)))
      tracking = (dev / env) < 0.05f;

   const bool update = tracking || pulseFilter > (uint32_t)(c.etu * 10);
   const float followed = env * c.envW0 + value * c.envW1;

   env = update ? followed : (clock < (uint32_t)c.etu ? value : env);
   pulseFilter = update ? 0u : pulseFilter;

   envelope = env;
}

/* The front end without its history stores: envelope (conditional EMA), DC removal, mean deviation, average and the
 * carrier-edge peak tracker. It depends on nothing but the samples (no detector or lock state), which is what lets the
 * scan kernel (nfc_scan.hpp) run it ahead of the decoder. The caller has already advanced s.clock and s.pulseFilter. */
NFC_DEV NfcNow nfc_front_end_core(const NfcConfig &c, NfcStreamState &s, float value)
{
   nfc_envelope_step(c, s.clock, s.pulseFilter, s.env, value);

   float n0 = value + s.n1 * c.iirA;
   float filtered = n0 - s.n1;
   s.n1 = n0;

   s.mdev = s.mdev * c.mdevW0 + nfc_abs(filtered) * c.mdevW1;
   s.avg = s.avg * c.meanW0 + value * c.meanW1;

   NfcNow now;
   now.x = value;
   now.filt = filtered;
   now.mdev = s.mdev;
   now.depth = 0.0f;

   /* edge-peak tracker (selects: the three cases are mutually exclusive) */
   const float rectified = nfc_abs(filtered);
   const bool high = rectified > c.highThreshold;
   const bool peak = high && rectified > s.edgePeak;
   const bool low = !high && rectified < c.lowThreshold;

   s.edgeTime = peak ? s.clock : s.edgeTime;
   s.edgePeak = peak ? rectified : (low ? 0.0f : s.edgePeak);

   return now;
}

/* the caller has already advanced s.clock and s.pulseFilter for this sample */
NFC_DEV NfcNow nfc_front_end(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float value)
{
   NfcNow now = nfc_front_end_core(c, s, value);

   const float env = s.env;
   const float clamped = (value < 0.0f) ? 0.0f : ((env < value) ? env : value);

   now.depth = (env - clamped) / env;

   const uint32_t slot = s.clock & NFC_HMASK;

   NFC_AT(mem, NFC_R_X, slot) = now.x;
   NFC_AT(mem, NFC_R_FILT, s.clock & NFC_FMASK) = now.filt;
   NFC_AT(mem, NFC_R_MDEV, s.clock & NFC_FMASK) = now.mdev;
   NFC_AT(mem, NFC_R_DEPTH, s.clock & NFC_FMASK) = now.depth;

   return now;
}

/* ring positions idx % period for the correlators (idx = 1024 - delay + clock, as the reference's
 * offsetSignalIndex + signalClock). Incremental, except in a window around the 32-bit clock wrap and at
 * stream start where the exact modulo is taken. */
NFC_DEV bool nfc_exact_zone(uint32_t clock)
{
   return (uint32_t)(clock + 1024u) < 2048u;
}

NFC_DEV uint32_t nfc_bump(uint32_t pos, uint32_t period)
{
   ++pos;
   return pos >= period ? 0u : pos;
}

NFC_DEV uint32_t nfc_next_pos(uint32_t clock, bool exact, uint32_t pos, const NfcRate &rt, uint32_t period, uint32_t label)
{
   return exact ? ((uint32_t)(1024u - rt.delay + clock) % period + label) % period : nfc_bump(pos, period);
}

/* Ring phase label of the correlation ring that starts at `base` (NfcStreamCold::label): zero for a stream that has
 * only ever been decoded sequentially; a lane of the time-parallel path numbers its rings from its own first sample.
 * Only the exact-modulo variants need it (the common ones advance whatever positions they are given). */
NFC_DEV uint32_t nfc_ring_label(const NfcLaneMem &mem, uint32_t base, uint32_t period)
{
   const NfcConfig &c = *mem.tables;
   uint32_t k = 0;

   for (uint32_t i = 1; i < 6; i++)
      k = base >= c.corrOffset[i] ? i : k;

   if (k == 5 && period == c.v.p0)
      k = 6;

   return mem.cold->label[k];
}

NFC_DEV void nfc_advance_positions(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem)
{
   const bool exact = mem.exact;

   /* written out per correlator: every state field keeps a compile-time address, so the record stays in VGPRs */
   const uint32_t *label = mem.cold->label; /* read by the exact variants only */

   s.posA[0] = nfc_next_pos(s.clock, exact, s.posA[0], c.a[0], c.a[0].p1, exact ? label[0] : 0u);
   s.posA[1] = nfc_next_pos(s.clock, exact, s.posA[1], c.a[1], c.a[1].p1, exact ? label[1] : 0u);
   s.posA[2] = nfc_next_pos(s.clock, exact, s.posA[2], c.a[2], c.a[2].p1, exact ? label[2] : 0u);
   s.posF[0] = nfc_next_pos(s.clock, exact, s.posF[0], c.f[1], c.f[1].p1, exact ? label[3] : 0u);
   s.posF[1] = nfc_next_pos(s.clock, exact, s.posF[1], c.f[2], c.f[2].p1, exact ? label[4] : 0u);
   s.posV1 = nfc_next_pos(s.clock, exact, s.posV1, c.v, c.v.p1, exact ? label[5] : 0u);
   s.posV0 = nfc_next_pos(s.clock, exact, s.posV0, c.v, c.v.p0, exact ? label[6] : 0u);
}

/* ring position of the locked correlator: a private copy taken at lock time and advanced alongside the others
 * (selecting among posA/posF by index would put the whole state record back into scratch memory) */
NFC_DEV void nfc_advance_lock_pos(NfcStreamState &s, const NfcLaneMem &mem)
{
   if (mem.exact)
      s.u.decode.lockPos = ((uint32_t)(1024u - s.u.decode.rt.delay + s.clock) % s.u.decode.rt.p1 +
                            nfc_ring_label(mem, s.u.decode.lockBase, s.u.decode.rt.p1)) % s.u.decode.rt.p1;
   else
      s.u.decode.lockPos = nfc_bump(s.u.decode.lockPos, s.u.decode.rt.p1);
}

NFC_DEV uint32_t nfc_lock_pos(const NfcStreamState &s)
{
   return s.u.decode.lockPos;
}

/* (idx + add) % period given pos = idx % period; exact modulo near the clock wrap */
NFC_DEV uint32_t nfc_point(const NfcLaneMem &mem, uint32_t clock, uint32_t delay, uint32_t pos, uint32_t add, uint32_t period, uint32_t base)
{
   if (mem.exact)
      return ((uint32_t)(1024u - delay + clock + add) % period + nfc_ring_label(mem, base, period)) % period;

   uint32_t p = pos + add;
   return p >= period ? p - period : p;
}

/* carrier presence, NfcDecoder.cpp:472-523 */
NFC_DEV void nfc_detect_carrier(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem)
{
   if (s.avg > c.highThreshold)
   {
      if (!s.carrierOn)
      {
         s.carrierOn = s.edgeTime ? s.edgeTime : s.clock;
         nfc_emit(mem, s, NFC_TECH_ANY, NFC_FRAME_CARRIER_ON, 0, NFC_PHASE_CARRIER, 0, s.carrierOn, s.carrierOn, nullptr, 0);
         s.carrierOff = 0;
         s.edgeTime = 0;
         mem.cold->emitClock = s.clock;
         mem.cold->emitValid = 1;
         mem.cold->emitOwn = 1;
      }
   }
   else if (s.avg < c.lowThreshold)
   {
      if (!s.carrierOff)
      {
         s.carrierOff = s.edgeTime ? s.edgeTime : s.clock;
         nfc_emit(mem, s, NFC_TECH_ANY, NFC_FRAME_CARRIER_OFF, 0, NFC_PHASE_CARRIER, 0, s.carrierOff, s.carrierOff, nullptr, 0);
         s.carrierOn = 0;
         s.edgeTime = 0;
         mem.cold->emitClock = s.clock;
         mem.cold->emitValid = 1;
         mem.cold->emitOwn = 1;
      }
   }
}

/* ------------------------------------------------------------------------------------------ */
/* the shared sliding correlator: box sum of width p2 over the delayed signal, kept  */
/* in a p1-deep ring; S0/S1 are differences of ring entries. Memory reads ("taps") are split from  */
/* the arithmetic so that a step can issue all of its reads first and pay one memory latency.     */
/* ------------------------------------------------------------------------------------------ */

struct NfcTap
{
   float in;   /* value entering the window */
   float out;  /* value leaving the window  */
   float c2;   /* ring[(idx + p2) % p1]     */
   float c3;   /* ring[(idx - 1) % p1]      */
};

struct NfcCorr
{
   float s0, s1;
};

/* taps of the raw-signal correlator; `fresh` = the entering value is the current sample (delay 0) */
NFC_DEV NfcTap nfc_tap_raw(const NfcLaneMem &mem, uint32_t clock, const NfcRate &rt, uint32_t base, uint32_t pos, bool needC3)
{
   NfcTap t;
   const uint32_t cur = clock - rt.delay;
   t.in = NFC_AT(mem, NFC_R_X, cur & NFC_HMASK);
   t.out = NFC_AT(mem, 0u, NFC_X_OLD_INDEX(mem, cur - rt.p2));
   t.c2 = NFC_AT(mem, NFC_R_CORR, base + nfc_point(mem, clock, rt.delay, pos, rt.p2, rt.p1, base));
   t.c3 = needC3 ? NFC_AT(mem, NFC_R_CORR, base + nfc_point(mem, clock, rt.delay, pos, rt.p1 - 1u, rt.p1, base)) : 0.0f;
   return t;
}

/* ring[(idx - 1) % p1] for a search-mode correlator: when the detector bank stepped on the previous sample it is
 * exactly the running sum before this sample's update (that is what was stored there one sample ago, and resets
 * clear ring and sum together); only after a gap in the search does it have to be read back */
template <class M>
NFC_DEV float nfc_previous_sum(const NfcLaneMem &mem, const NfcStreamState &s, const M &m, const NfcRate &rt, uint32_t base, uint32_t pos)
{
   if (s.bankClock == s.clock - 1u)
      return m.acc;

   /* rare (first search sample after a gap). The value is marked as used here so that the wait for this load sits in
    * this branch: otherwise the compiler waits where the two paths meet, on every sample, and with an in-order memory
    * counter that wait also covers the ring store of the detector before this one (a full store round trip per
    * correlator per sample: it was 40 % of the idle step) */
   float previous = NFC_AT(mem, NFC_R_CORR, base + nfc_point(mem, s.clock, rt.delay, pos, rt.p1 - 1u, rt.p1, base));
   NFC_ARRIVED(previous);
   return previous;
}

template <class M>
NFC_DEV NfcCorr nfc_corr_apply(const NfcLaneMem &mem, M &m, const NfcTap &t, uint32_t base, uint32_t pos)
{
   m.acc += t.in;
   m.acc -= t.out;

   NFC_AT(mem, NFC_R_CORR, base + pos) = m.acc;

   NfcCorr r;
   r.s0 = m.acc - t.c2;
   r.s1 = t.c2 - t.c3;
   return r;
}

/* same correlator over 10*filtered^2 (listen ASK, NfcA.cpp:1115-1131); window w = p2 (NFC-A) */
NFC_DEV NfcCorr nfc_correlate_power(const NfcLaneMem &mem, uint32_t clock, NfcMod &m, const NfcRate &rt, uint32_t base, uint32_t pos,
                                    float v, float old, float c2, float c3)
{
   const uint32_t cur = clock - rt.delay;
   const float sq = v * v * 10.0f;

   NFC_AT(mem, NFC_R_PROD, cur & NFC_PMASK) = sq;

   m.acc += sq;
   m.acc -= old;

   NFC_AT(mem, NFC_R_CORR, base + pos) = m.acc;

   NfcCorr r;
   r.s0 = m.acc - c2;
   r.s1 = c2 - c3;
   return r;
}

/* delayed self product for BPSK listen frames (NfcA.cpp:1236-1244, NfcB.cpp:785-796); returns the product and
 * the product leaving the p4 window */
struct NfcPhase
{
   float in, out;
};

NFC_DEV NfcPhase nfc_phase_product(const NfcLaneMem &mem, uint32_t clock, const NfcRate &rt, float a, float b, float leaving)
{
   const uint32_t cur = clock - rt.delay;
   NfcPhase p;
   p.out = leaving;
   p.in = a * b * 10.0f;
   NFC_AT(mem, NFC_R_PROD, cur & NFC_PMASK) = p.in;
   return p;
}

NFC_DEV void nfc_phase_integrate(NfcMod &m, const NfcPhase &p)
{
   m.phaseAcc += p.in;
   m.phaseAcc -= p.out;
}

/* A detector that recognises its start-of-frame prepares the decode register set in HBM (NfcStreamCold::init):
 * everything zero, then what the detector found (the reference's record at that point holds exactly those values,
 * see nfc_types.h). Installing it happens once per search step, in nfc_enter_lock: eight inlined copies of a
 * 120-register state swap would otherwise surround the hot path with register shuffling. */
NFC_DEV NfcDecodeRegs &nfc_take_lock(const NfcLaneMem &mem, const NfcRate &rt, uint32_t rate, uint32_t base, uint32_t pos)
{
   NfcDecodeRegs &d = mem.cold->init;

   nfc_mod_clear(d.lock);
   d.rt = rt;
   d.lockRate = rate;
   d.pulseCode = 0;
   d.lockBase = base;
   d.lockPos = pos;
   d.guardEnd = 0;
   d.waitingEnd = 0;
   d.symPattern = 0; d.symValue = 0; d.symStart = 0; d.symEnd = 0; d.symEdge = 0;
   d.bsPrevious = 0; d.bsBits = 0; d.bsSkip = 0; d.bsData = 0; d.bsFlags = 0; d.bsParity = 0; d.bsBytes = 0;
   d.frameType = 0; d.frameRate = 0; d.frameStart = 0; d.frameEnd = 0;
   d.maxFrame = 0;
   d.pendType = 0;
   d.pendFlags = 0;

   return d;
}

/* enter decode mode: park the detector records (they stay frozen while locked) and install the prepared decode set */
NFC_DEV void nfc_enter_lock(NfcStreamState &s, const NfcLaneMem &mem, uint32_t tech)
{
   mem.cold->parked = s.u.search;
   s.u.decode = mem.cold->init;
   s.u.decode.maxFrame = mem.cold->tim[tech - NFC_TECH_A].maxFrameSize;
   if (mem.linked)
      *mem.flags |= 1u << (tech - NFC_TECH_A);
   s.lockTech = tech;
   NFC_DRAIN();
}

/* leave decode mode (the technology resets). The decode register set dies here; bringing the detector records
 * back, clearing those of the technology that was locked and zeroing its rings (the reference's resetModulation)
 * happens once, at the end of the decode step (nfc_finish_unlock), not at each of the many reset sites */
NFC_DEV void nfc_leave_lock(NfcStreamState &s, uint32_t tech)
{
   s.lockTech = 0;
   s.unlock = tech;
}

/* a/w compared against +-limit: the IEEE division is only needed when |a| is within 0.1 % of w*limit or beyond
 * it. If |a| <= 0.999*w*limit then |fl(a/w)| <= limit*0.999*(1+2^-23) < limit, so every `> limit` / `< -limit`
 * test on the quotient is false and the quotient itself is never used; NaNs take the skip path on both sides. */
NFC_DEV bool nfc_may_exceed(float a, float w, float limit)
{
   return nfc_abs(a) > w * limit * 0.999f;
}

/* History reads of one decode-mode step, whatever the locked technology and frame stage: issued together before
 * the front end stores the new sample so that the step pays one memory latency instead of one per (divergent)
 * decode path. Which of them a path uses depends on its stage; unused ones are harmless (addresses always valid). */
struct NfcDecTaps
{
   float x0, x2; /* raw signal at the (delayed) decode point and half a symbol before        */
   float f0, f1; /* DC-removed signal at the decode point and one symbol before               */
   float m0, d0; /* mean deviation / modulation depth at the decode point                     */
   float pp;     /* product ring entry leaving the listen-mode integration window             */
   float c2, c3; /* correlation ring entries half a symbol (V listen: one symbol) / one sample back */
};

/* The decode-mode reads are issued for the whole block (nfc_step_as). What a lane really needs depends on its frame
 * stage: the raw-signal correlators (poll frames of NFC-A/F/V, all of NFC-F) read x0 x2 c2 c3, the power / phase
 * correlators (listen frames of NFC-A/B/V) f0 f1 pp c2 c3, NFC-B poll frames f0 d0; with no detection delay the four
 * values at the decode point come from the front end, not from memory. A read a lane does not need (and every read
 * of a lane that is not locked: `valid` false, its decode registers hold detector records) is pointed at the lane's
 * correlation-ring entry, a row the block fetches anyway, and a read no lane of the block needs is not issued. */
NFC_DEV NfcDecTaps nfc_load_decode_taps(const NfcLaneMem &mem, const NfcStreamState &s, bool valid)
{
   const NfcDecodeRegs &d = s.u.decode;
   const NfcRate &rt = d.rt;
   const uint32_t cur = s.clock - rt.delay;
   const bool vListen = (s.lockTech == NFC_TECH_V) && (d.frameType == NFC_FRAME_LISTEN);

   /* integration window of the listen-mode product ring: p2 for NFC-A 106k (ASK), p1 for NFC-V, p4 for BPSK */
   const uint32_t window = (s.lockTech == NFC_TECH_A && d.lockRate == 0) ? rt.p2 : ((s.lockTech == NFC_TECH_V) ? rt.p1 : rt.p4);

   const bool poll = d.frameType == NFC_FRAME_POLL;
   const bool raw = valid && (s.lockTech == NFC_TECH_F || (poll && s.lockTech != NFC_TECH_B));
   const bool filt = valid && !raw;
   const bool stored = valid && rt.delay != 0; /* the decode point lies in the past: its values are in the rings */

   const uint32_t p2 = vListen ? nfc_point(mem, s.clock, rt.delay, s.posV0, rt.p1, rt.p0, d.lockBase)
                               : nfc_point(mem, s.clock, rt.delay, d.lockPos, rt.p2, rt.p1, d.lockBase);
   const uint32_t p3 = nfc_point(mem, s.clock, rt.delay, d.lockPos, rt.p1 - 1u, rt.p1, d.lockBase);

   /* slot index (within the stream block) of the row every lane reads anyway */
   const uint32_t common = NFC_R_CORR + (valid ? d.lockBase + p2 : 0u);

   NfcDecTaps t;

   t.c2 = NFC_AT(mem, 0u, common);
   t.c3 = NFC_AT(mem, 0u, valid ? NFC_R_CORR + d.lockBase + p3 : common);

   t.x0 = t.x2 = t.f0 = t.f1 = t.m0 = t.d0 = t.pp = 0.0f;

   if (NFC_ANY(raw && stored))
      t.x0 = NFC_AT(mem, 0u, raw && stored ? NFC_R_X + (cur & NFC_HMASK) : common);
   if (NFC_ANY(raw))
      t.x2 = NFC_AT(mem, 0u, raw ? NFC_X_OLD_INDEX(mem, cur - rt.p2) : common);
   if (NFC_ANY(filt && stored))
      t.f0 = NFC_F_TAP(mem, filt && stored, NFC_R_FILT, cur, common);
   if (NFC_ANY(filt))
   {
      /* (one symbol further back: BPSK only, NFC-A / -B, at most 188 samples) */
      t.f1 = NFC_AT(mem, 0u, filt ? NFC_R_FILT + ((cur - rt.p1) & NFC_FMASK) : common);
      t.pp = NFC_AT(mem, 0u, filt ? NFC_R_PROD + ((cur - window) & NFC_PMASK) : common);
   }
   if (NFC_ANY(stored))
      t.m0 = NFC_F_TAP(mem, stored, NFC_R_MDEV, cur, common);
   if (NFC_ANY(filt && stored && poll))
      t.d0 = NFC_F_TAP(mem, filt && stored && poll, NFC_R_DEPTH, cur, common);

   return t;
}

/* a lane of the time-parallel path notes that it has set lastCommand of technology k (NfcStreamCold::usedTech) */
NFC_DEV void nfc_command_written(const NfcLaneMem &mem, uint32_t k)
{
   if (mem.linked)
      *mem.flags |= 1u << (8u + k);
}

/* ---- what a lane of the time-parallel path requires of a waiting time it inherited (NfcStreamCold::waitUsed) ---- */

/* nfc*_process of a poll frame has just taken the waiting time from the technology's protoWaitingTime */
NFC_DEV void nfc_wait_from_proto(const NfcLaneMem &mem, uint32_t k)
{
   if (mem.linked)
   {
      const uint32_t f = mem.cold->waitFlags;
      mem.cold->waitFlags = (f & ~(1u << k)) | (((f >> (4u + k)) & 1u) ? 0u : (1u << k));
   }
}

/* ... and replaced it by a constant of the command at hand (REQA, SELECT, RATS, REQB, ATTRIB, REQC) */
NFC_DEV void nfc_wait_overridden(const NfcLaneMem &mem, uint32_t k)
{
   if (mem.linked)
      mem.cold->waitFlags &= ~(1u << k);
}

/* the lane has set protoWaitingTime of technology k */
NFC_DEV void nfc_wait_proto_written(const NfcLaneMem &mem, uint32_t k)
{
   if (mem.linked)
      mem.cold->waitFlags |= 1u << (4u + k);
}

/* The search for the start of an answer of technology k has ended on this sample: by the time running out (ranOut: the
 * comparison `clock > waitingEnd` itself), or by anything else - a start of frame, a modulation deeper than a card's - with
 * the comparison false up to and including this sample. */
NFC_DEV void nfc_wait_ended(const NfcLaneMem &mem, const NfcStreamState &s, uint32_t k, bool ranOut)
{
   if (!mem.linked)
      return;

   const uint32_t f = mem.cold->waitFlags;

   if (!((f >> k) & 1u))
      return; /* (a waiting time of the command's own, or one the lane has set itself) */

   if (ranOut)
      mem.cold->waitFlags = (f & ~(1u << k)) | (1u << (8u + k));
   else
   {
      /* waitingEnd = (sample the waiting time counts from) + waitingTime: what of it had passed by now */
      const uint32_t used = mem.cold->tim[k].waitingTime - (s.u.decode.waitingEnd - s.clock);

      if (used > mem.cold->waitUsed[k])
         mem.cold->waitUsed[k] = used;

      mem.cold->waitFlags = f & ~(1u << k);
   }
}

/* a frame has been assembled on this sample: remember it; classification (process*), emission and the mode change
 * that follows are done in one place per decode step */
NFC_DEV void nfc_pend_frame(NfcStreamState &s, uint32_t type, uint32_t flags)
{
   s.u.decode.pendType = type;
   s.u.decode.pendFlags = flags;
}

#include "nfc_tech_a.hpp"
#include "nfc_tech_b.hpp"
#include "nfc_tech_f.hpp"
#include "nfc_tech_v.hpp"

/* ------------------------------------------------------------------------------------------ */
/* one sample                                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* Search mode (no technology locked), NfcDecoder.cpp:394-418: the detector bank on the sample the front end has just
 * produced; the first detector that recognises its start of frame wins, later ones skip this sample. */
NFC_DEV void nfc_search_detect(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now,
                               const NfcTapsA &ta, const NfcTapsB &tb, const NfcTapsF &tf, const NfcTapsV &tv)
{
   nfc_detect_carrier(c, s, mem);

   /* the detectors run once the decoder has seen 1024 samples and while there is a carrier (the gates at the top of
    * each detectModulation: NfcA.cpp:220-225, NfcB.cpp:241-246, NfcF.cpp:209-214, NfcV.cpp:239-244) */
   const bool armed = s.clock >= 1024u && !(s.env < c.powerThreshold);

   uint32_t locked = 0;

   if (armed)
   {
      /* first detector that recognises its start of frame wins, later ones skip this sample */
      if ((c.enabled & 1u) && nfca_detect(c, s, mem, ta, now))
         locked = NFC_TECH_A;
      else if ((c.enabled & 2u) && nfcb_detect(c, s, mem, tb, now))
         locked = NFC_TECH_B;
      else if ((c.enabled & 4u) && nfcf_detect(c, s, mem, tf, now))
         locked = NFC_TECH_F;
      else if ((c.enabled & 8u) && nfcv_detect(c, s, mem, tv, now))
         locked = NFC_TECH_V;

      /* every detector that is enabled stepped its correlator on this sample: on the next sample
       * ring[(idx - 1) % p1] is known to equal the running sum and need not be read back */
      if (!locked)
      {
         if (s.bankClock != s.clock - 1u)
            mem.cold->bankRun = s.clock; /* first step after a gap (rare): where the unbroken run of the bank begins */
         s.bankClock = s.clock;
      }
   }

   if (locked)
      nfc_enter_lock(s, mem, locked);
}

/* the tail of the reference's decodePollFrame / decodeListenFrame for all four technologies: build the frame, run the
 * protocol processing (which feeds the timing back), emit it, then either prepare the listen window (poll frames:
 * "clear modulation status for receiving card response") or fall back to search (listen frames: resetModulation) */
NFC_DEV void nfc_finish_frame(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem)
{
   NfcDecodeRegs &d = s.u.decode;

   const uint32_t tech = s.lockTech;
   const uint32_t type = d.pendType;
   uint32_t flags = d.pendFlags;
   uint32_t phase = 0;

   d.pendType = 0;
   d.pendFlags = 0;

   const uint8_t *data = mem.bytes;
   uint32_t len = d.bsBytes > NFC_STREAM_BYTES ? NFC_STREAM_BYTES : d.bsBytes;

   if (tech == NFC_TECH_F)
   {
      /* the two synchronisation bytes are checked and stripped (NfcF.cpp:467-472) */
      if (mem.bytes[0] != 0xB2 || mem.bytes[1] != 0x4D)
         flags |= NFC_FLAG_SYNC;

      data += 2;
      len -= 2;
   }

   const uint32_t start = d.frameStart, end = d.frameEnd, rate = d.frameRate;

   /* ring of the locked correlator, for the poll-end clearing */
   const uint32_t ringBase = d.lockBase;
   const uint32_t ringSlots = tech == NFC_TECH_B ? 0u : (tech == NFC_TECH_V ? d.rt.p0 : d.rt.p1);

   uint32_t isA = tech, isB = tech, isF = tech, isV = tech;
   NFC_OPAQUE(isA);
   NFC_OPAQUE(isB);
   NFC_OPAQUE(isF);
   NFC_OPAQUE(isV);

   /* for the time-parallel path: did the classification read a lastCommand the lane had inherited? (NfcStreamCold::usedTech) */
   const uint32_t techIndex = tech - NFC_TECH_A;

   if (mem.linked)
   {
      const uint32_t seen = *mem.flags;

      /* a listen frame is classified by lastCommand: the lane's own, or the one it inherited */
      if (type != NFC_FRAME_POLL && !((seen >> (8u + techIndex)) & 1u))
         *mem.flags |= 1u << (4u + techIndex);

      /* The first frame of this technology the lane processes is the command that starts its protocol over - REQA / WUPA
       * (NfcA.cpp:1480-1510), REQB / WUPB, REQC - : it sets every field of the technology's timing, the times of the
       * frame at hand included, and (NFC-A) the chaining flags before anything reads them; what the lane had inherited
       * there is gone without having mattered. (The frame size limit read at the lock cannot have mattered either: the
       * smallest limit is 16 bytes, these frames are shorter.) */
      if (type == NFC_FRAME_POLL && !((seen >> (18u + techIndex)) & 1u))
      {
         const uint32_t b0 = nfc_byte(data, len, 0);
         const bool restart = (tech == NFC_TECH_A && len == 1 && (b0 == 0x26 || b0 == 0x52)) || (tech == NFC_TECH_B && len == 5 && b0 == 0x05) ||
                              (tech == NFC_TECH_F && nfc_byte(data, len, 1) == 0x00);
         if (restart)
            *mem.flags |= 1u << (22u + techIndex);
      }

      *mem.flags |= 1u << (18u + techIndex);
   }

   if (isA == NFC_TECH_A)
      nfca_process(c, s, mem, type, data, len, flags, phase);
   if (isB == NFC_TECH_B)
      nfcb_process(c, s, mem, type, data, len, flags, phase);
   if (isF == NFC_TECH_F)
      nfcf_process(c, s, mem, type, data, len, flags, phase);
   if (isV == NFC_TECH_V)
      nfcv_process(c, s, mem, type, data, len, flags, phase);


   nfc_emit(mem, s, tech, type, flags, phase, rate, start, end, data, len);

   /* NFC-A HLTA resets inside process() */
   const bool listenNext = type == NFC_FRAME_POLL && s.lockTech == tech;
   const bool searchNext = type != NFC_FRAME_POLL;

   if (listenNext)
   {
      nfc_clear_assembly(s);
      nfc_poll_end_clear(mem, d.lock, ringBase, ringSlots);
   }

   if (searchNext)
      nfc_leave_lock(s, tech);
}

NFC_DEV void nfc_finish_unlock(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem)
{
   const uint32_t tech = s.unlock;

   s.unlock = 0;
   s.u.search = mem.cold->parked;
   mem.cold->lastUnlock = s.clock;

   uint32_t isA = tech, isB = tech, isF = tech, isV = tech;
   NFC_OPAQUE(isA);
   NFC_OPAQUE(isB);
   NFC_OPAQUE(isF);
   NFC_OPAQUE(isV);

   if (isA == NFC_TECH_A)
   {
      nfc_mod_clear(s.u.search.detA[0]);
      nfc_mod_clear(s.u.search.detA[1]);
      nfc_mod_clear(s.u.search.detA[2]);
      /* the three rings are adjacent */
      nfc_zero_ring(mem, NFC_R_CORR + c.corrOffset[0], c.a[0].p1 + c.a[1].p1 + c.a[2].p1);
   }

   if (isB == NFC_TECH_B)
   {
      nfc_mod_clear(s.u.search.detB[0]);
      nfc_mod_clear(s.u.search.detB[1]);
   }

   if (isF == NFC_TECH_F)
   {
      nfc_mod_clear(s.u.search.detF[0]);
      nfc_mod_clear(s.u.search.detF[1]);
      mem.cold->clearedF[0] = 1;
      mem.cold->clearedF[1] = 1;
      mem.cold->boundF[0].flags |= NFC_FBOUND_THR_OWN;
      mem.cold->boundF[1].flags |= NFC_FBOUND_THR_OWN;
      /* the two rings are adjacent */
      nfc_zero_ring(mem, NFC_R_CORR + c.corrOffset[3], c.f[1].p1 + c.f[2].p1);
   }

   if (isV == NFC_TECH_V)
   {
      nfc_mod_clear(s.u.search.detV);
      nfc_zero_ring(mem, NFC_R_CORR + c.corrOffset[5], c.v.p0);
   }

   NFC_DRAIN();
}

/* One sample. The mode the sample is handled in is the one the stream is in when the sample arrives (a detector
 * that locks, or a frame end that unlocks, takes effect with the next sample, as in the reference where the locked
 * decoder is entered / left after the sample that decided it). `exact` is wave-uniform and must be true whenever
 * nfc_exact_zone(s.clock + 1) is for any lane (it may be true more often: the exact ring positions are always
 * right, only slower to obtain).
 *
 * All history reads of the step (the eight detectors' in search mode, the locked correlator's in decode mode) are
 * issued before the front end stores this sample: none of them can alias the slot being written (their delays are
 * > 0, or the value is patched in below), so a step pays one memory latency. */
/* what a caller that has the front end's results already (the wave decoder: computed ahead for the whole submission,
 * nfc_scan.h) hands to the step instead of the raw sample; the history rings then hold the sample already */
struct NfcGiven
{
   NfcNow now;
   float env; /* signalEnvelope after this sample */
   float avg; /* signalAverage after this sample */
};

template <bool EXACT, bool GIVEN>
NFC_DEV void nfc_step_impl(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &lane, float value, const NfcGiven *given)
{
   NfcLaneMem mem = lane;
   mem.exact = EXACT;

   ++s.clock;
   if (!GIVEN)
      ++s.pulseFilter; /* (the envelope tracker's counter: with the front end given it is not kept) */

   nfc_advance_positions(c, s, mem);

   const uint32_t mode = s.lockTech;

   uint32_t inSearch = mode, inDecode = mode;
   NFC_OPAQUE(inSearch);
   NFC_OPAQUE(inDecode);

   /* History reads of the step, issued for the whole block under wave-uniform conditions (values produced inside a
    * divergent region and consumed in a later one would be waited for at the end of the first), all before the front
    * end stores the sample: one memory latency per step whatever mix of modes the block is in. */
   NfcDecTaps taps;
   NfcTapsA ta;
   NfcTapsB tb;
   NfcTapsF tf;
   NfcTapsV tv;

   uint32_t locked = mode;
   NFC_OPAQUE(locked);

   if (locked != 0)
      nfc_advance_lock_pos(s, mem);

   if (NFC_ANY(mode != 0))
      taps = nfc_load_decode_taps(mem, s, mode != 0);

   if (NFC_ANY(mode == 0))
   {
      /* the addresses are always inside the stream block, and a detector that is disabled or not yet armed (first
       * 1024 samples) simply ignores what was read */
      nfca_load_taps(c, s, mem, ta);
      nfcb_load_taps(c, s, mem, tb);
      nfcf_load_taps(c, s, mem, tf);
      nfcv_load_taps(c, s, mem, tv);
   }

   /* the front end is the same in both modes */
   NfcNow now;

   if (GIVEN)
   {
      now = given->now;
      s.env = given->env;
      s.avg = given->avg;
      s.mdev = now.mdev;
   }
   else
      now = nfc_front_end(c, s, mem, value);

   if (inSearch == 0)
      nfc_search_detect(c, s, mem, now, ta, tb, tf, tv);

   if (inDecode != 0)
   {
      /* without delay the decode point is the sample the front end has just produced (not in memory when the taps
       * were read) */
      if (s.u.decode.rt.delay == 0)
      {
         taps.x0 = now.x;
         taps.f0 = now.filt;
         taps.m0 = now.mdev;
         taps.d0 = now.depth;
      }

      uint32_t isA = mode, isB = mode, isF = mode, isV = mode;
      NFC_OPAQUE(isA);
      NFC_OPAQUE(isB);
      NFC_OPAQUE(isF);
      NFC_OPAQUE(isV);

      if (isA == NFC_TECH_A)
         nfca_decode(c, s, mem, now, taps);

      if (isB == NFC_TECH_B)
         nfcb_decode(c, s, mem, now, taps);

      if (isF == NFC_TECH_F)
         nfcf_decode(c, s, mem, now, taps);

      if (isV == NFC_TECH_V)
         nfcv_decode(c, s, mem, now, taps);
   }

   uint32_t pending = mode ? s.u.decode.pendType : 0u;
   NFC_OPAQUE(pending);

   if (pending)
      nfc_finish_frame(c, s, mem);

   if (s.unlock)
      nfc_finish_unlock(c, s, mem);
}

template <bool EXACT>
NFC_DEV void nfc_step_as(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &lane, float value)
{
   nfc_step_impl<EXACT, false>(c, s, lane, value, nullptr);
}

/* ------------------------------------------------------------------------------------------ */
/* lanes of the time-parallel path (nfc_scan.h): warm-up steps and the "at rest" test           */
/* ------------------------------------------------------------------------------------------ */

/* Front end only: the first part of a window lane's warm-up refills the sample history (x, filtered, deviation, depth)
 * its detectors and decoders look back into. */
template <bool EXACT>
NFC_DEV void nfc_step_front(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &lane, float value)
{
   NfcLaneMem mem = lane;
   mem.exact = EXACT;

   ++s.clock;
   ++s.pulseFilter;

   nfc_advance_positions(c, s, mem);
   (void)nfc_front_end(c, s, mem, value);
}

/* Front end + upkeep of the six search correlators (running box sum and ring entry), no decisions: the second part of
 * the warm-up. The sums start from zero instead of from the reference's value at that sample; every use of them is a
 * difference of two ring entries or of an entry and the running sum (S0, S1, nfc_corr_apply), and on the int16 grid all
 * of these are exact, so a constant offset never shows (nfc_scan.h). */
template <bool EXACT, bool GIVEN = false>
NFC_DEV void nfc_step_upkeep(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &lane, float value, const NfcGiven *given = nullptr)
{
   NfcLaneMem mem = lane;
   mem.exact = EXACT;

   ++s.clock;
   if (!GIVEN)
      ++s.pulseFilter;

   nfc_advance_positions(c, s, mem);

   NfcTapsA ta;
   NfcTapsF tf;
   NfcTapsV tv;

   nfca_load_taps(c, s, mem, ta);
   nfcf_load_taps(c, s, mem, tf);
   nfcv_load_taps(c, s, mem, tv);

   NfcNow now;

   if (GIVEN)
   {
      now = given->now;
      s.env = given->env;
      s.avg = given->avg;
      s.mdev = now.mdev;
   }
   else
      now = nfc_front_end(c, s, mem, value);

   NfcSearchRegs &r = s.u.search;

   r.detA[0].acc += c.a[0].delay ? ta.t[0].in : now.x;
   r.detA[0].acc -= ta.t[0].out;
   NFC_AT(mem, NFC_R_CORR, c.corrOffset[0] + s.posA[0]) = r.detA[0].acc;

   r.detA[1].acc += c.a[1].delay ? ta.t[1].in : now.x;
   r.detA[1].acc -= ta.t[1].out;
   NFC_AT(mem, NFC_R_CORR, c.corrOffset[1] + s.posA[1]) = r.detA[1].acc;

   r.detA[2].acc += c.a[2].delay ? ta.t[2].in : now.x;
   r.detA[2].acc -= ta.t[2].out;
   NFC_AT(mem, NFC_R_CORR, c.corrOffset[2] + s.posA[2]) = r.detA[2].acc;

   r.detF[0].acc += now.x;
   r.detF[0].acc -= tf.t[1].out;
   NFC_AT(mem, NFC_R_CORR, c.corrOffset[3] + s.posF[0]) = r.detF[0].acc;

   r.detF[1].acc += now.x;
   r.detF[1].acc -= tf.t[2].out;
   NFC_AT(mem, NFC_R_CORR, c.corrOffset[4] + s.posF[1]) = r.detF[1].acc;

   r.detV.acc += tv.t.in;
   r.detV.acc -= tv.t.out;
   NFC_AT(mem, NFC_R_CORR, c.corrOffset[5] + s.posV1) = r.detV.acc;

   if (s.bankClock != s.clock - 1u)
      mem.cold->bankRun = s.clock;
   s.bankClock = s.clock;
}

/* True when the decoder is searching and every detector record is what a cleared record is, apart from the running
 * sums, from fields that are rewritten before they are next read (NFC-B: thr, recomputed on every sample of the idle
 * detector; NFC-F: syncValue, c0, lastValue, lastPhase, set by the first pulse of the next preamble before they can
 * decide anything) and from the two NFC-F fields a partial reset leaves behind, which travel with the lane's carry
 * (NfcCarry::pulsesF / thrF). From such a state the next thing a detector does is decided by the signal alone. */
NFC_DEV bool nfc_at_rest(const NfcStreamState &s)
{
   const NfcSearchRegs &r = s.u.search;
   uint32_t busy = s.lockTech | s.unlock;

   for (int i = 0; i < 3; i++)
      busy |= r.detA[i].winStart | r.detA[i].winEnd | r.detA[i].symStart | r.detA[i].peakTime | nfc_bits(r.detA[i].peak) | nfc_bits(r.detA[i].aux);

   for (int i = 0; i < 2; i++)
      busy |= r.detB[i].winStart | r.detB[i].winEnd | r.detB[i].symStart | r.detB[i].symEnd | r.detB[i].auxTime | nfc_bits(r.detB[i].aux);

   for (int i = 0; i < 2; i++)
      busy |= r.detF[i].winStart | r.detF[i].winEnd | r.detF[i].sync | r.detF[i].symStart | r.detF[i].symEnd | nfc_bits(r.detF[i].peak) |
              r.detF[i].peakTime;

   busy |= r.detV.winStart | r.detV.winEnd | r.detV.symStart | r.detV.peakTime | nfc_bits(r.detV.peak) | nfc_bits(r.detV.aux);

   return busy == 0;
}

/* At rest, or with an NFC-F preamble detector that came back from another technology's lock with its windows in the
 * past (NfcF.cpp:262-283 only clears a record on a deep pulse or after a peak): such a record sits there until the next
 * pulse that exceeds the correlation threshold - no window end or synchronisation sample is ahead, no peak is waiting
 * for its timeout - so through quiet tiles it is as inert as a cleared one. It is not the same as a cleared one (the
 * next pulse finds `sync` set), which is why it travels with the lane's carry (NfcCarry::search) as it is. */
NFC_DEV bool nfc_quiescent(const NfcStreamState &s)
{
   const NfcSearchRegs &r = s.u.search;
   uint32_t busy = s.lockTech | s.unlock;

   for (int i = 0; i < 3; i++)
      busy |= r.detA[i].winStart | r.detA[i].winEnd | r.detA[i].symStart | r.detA[i].peakTime | nfc_bits(r.detA[i].peak) | nfc_bits(r.detA[i].aux);

   for (int i = 0; i < 2; i++)
      busy |= r.detB[i].winStart | r.detB[i].winEnd | r.detB[i].symStart | r.detB[i].symEnd | r.detB[i].auxTime | nfc_bits(r.detB[i].aux);

   busy |= r.detV.winStart | r.detV.winEnd | r.detV.symStart | r.detV.peakTime | nfc_bits(r.detV.peak) | nfc_bits(r.detV.aux);

   if (busy)
      return false;

   /* a time of the past: zero, or at least 256 samples (and less than 2^31) ago */
   auto past = [&](uint32_t t) { return t == 0u || (uint32_t)(s.clock - t - 256u) < 0x7FFFFF00u; };

   for (int i = 0; i < 2; i++)
   {
      const NfcDetF &f = r.detF[i];

      if (f.peakTime | nfc_bits(f.peak))
         return false;

      if (!past(f.winStart) || !past(f.winEnd) || !past(f.sync))
         return false;
   }

   return true;
}

/* run-time selection of the variant (CPU test build of this text; the kernels instantiate one variant each) */
NFC_DEV void nfc_step(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &lane, float value, bool exact)
{
   if (exact)
      nfc_step_as<true>(c, s, lane, value);
   else
      nfc_step_as<false>(c, s, lane, value);
}

/* state of a freshly initialised decoder; `keep` carries over what the reference's initialize()
 * leaves untouched (envelope / IIR / EMA scalars, carrier bookkeeping): NfcDecoder.cpp:295-360 */
NFC_DEV void nfc_state_init(const NfcConfig &c, NfcStreamState &s, NfcStreamCold &cold, bool keepFrontEnd)
{
   float env = s.env, n1 = s.n1, mdev = s.mdev, avg = s.avg, edgePeak = s.edgePeak;
   uint32_t pulse = s.pulseFilter, edgeTime = s.edgeTime, off = s.carrierOff, on = s.carrierOn;

   /* memset, not stores through a uint32_t alias: with type-based alias analysis the compiler treated the float
    * fields saved above as untouched by integer stores and dropped the `s.env = env` restores below as redundant,
    * so a re-initialised stream lost its envelope / filter / average on the GPU (round-1 hardware-only failure) */
   __builtin_memset(&s, 0, sizeof(NfcStreamState));
   __builtin_memset(&cold, 0, sizeof(NfcStreamCold));

   if (keepFrontEnd)
   {
      s.env = env; s.n1 = n1; s.mdev = mdev; s.avg = avg; s.edgePeak = edgePeak;
      s.pulseFilter = pulse; s.edgeTime = edgeTime; s.carrierOff = off; s.carrierOn = on;
   }

   s.clock = 0xFFFFFFFFu;

   nfca_protocol_defaults(c, cold.tim[0]);
   nfcb_protocol_defaults(c, cold.tim[1]);
   nfcf_protocol_defaults(c, cold.tim[2]);
   nfcv_protocol_defaults(c, cold.tim[3]);

   for (int t = 0; t < 4; t++)
   {
      cold.tim[t].guardTime = cold.tim[t].protoGuardTime;
      cold.tim[t].waitingTime = cold.tim[t].protoWaitingTime;
   }
}

#endif
