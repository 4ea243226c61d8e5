/*
 * nfc_wave.hpp — the wave decoder: one wavefront decodes one lane of work of the time-parallel path (nfc_scan.h: the
 * carry lane of a stream, a speculative window, the lane that regenerates a stream's final state), 64 consecutive
 * samples per step, one sample per lane.
 *
 * Why. The stream-parallel step machine (nfc_core.hpp: one lane per stream, rings in HBM) pays ~1000 instruction slots
 * and a round trip to its history rings per sample; a lane of the time-parallel path is one stream, so a pass costs its
 * longest lane at 2-3 us per sample. Here the wave is the lane:
 *
 *   - the front end (NfcTech.cpp:28-105) is not computed at all: the scan kernel's second walk has left its results per
 *     sample ({filtered, envelope, deviation, average}: NfcScanArgs::planes), 64 of them are one coalesced 1 KiB load;
 *   - one stream's history and correlation rings live in LDS (NfcWaveLds, ~21 KiB), the history 1024 deep so that a
 *     tile is written ahead of the sample at hand;
 *   - the decoder's state is one record in LDS (NfcWaveUni) that every lane reads; a sample that can change it is
 *     handled by the very step of nfc_core.hpp (nfc_step_impl<.., GIVEN>) run by the wave on a register copy of it, so
 *     whatever the reference does at that sample (NfcDecoder.cpp:394-418 and the tech decoders) is done by the same
 *     statement the stream-parallel kernels run;
 *   - samples that provably change nothing but the running sums and ring entries are not stepped: the wave evaluates
 *     the gate of the mode at hand for all remaining samples of the tile at once (nfc_wave_fast_*: correlations from
 *     wave prefix sums over the tile plus the ring as the tile found it, bit for bit the values the step would form),
 *     commits sums and ring entries of the uneventful run in bulk and steps the first eventful sample.
 *
 * Included by nfc_wave.hip (HIP, gfx950) and by tests/hostsim/emu_wave.cpp (test infrastructure: the same text run by 64
 * fibres per wave on the CPU). The includer provides, besides what nfc_core.hpp and nfc_scan.hpp ask for:
 *   NFC_WAVE_LANE()               lane 0..63
 *   NFC_WAVE_BARRIER()            LDS written before it is visible to every lane after it
 *   NFC_WAVE_BALLOT(p)            uint64_t
 *   NFC_WAVE_UNIFORM_BEGIN / NFC_WAVE_UNIFORM_END
 *                                 bracket code that works on the wave's shared records only (the device runs it in all
 *                                 lanes, which hold the same values; the fibre build runs it in lane 0)
 *   NFC_WAVE_UNIFORM_U32(x)       a value every lane holds, as a scalar
 *   NFC_WAVE_SCAN_ADD_F(v)        inclusive prefix sum over the lanes; NFC_WAVE_MAX_F(v): maximum, in every lane
 *   NFC_WAVE_PICK_F(reg, array, j)  element j (uniform) of a per-lane value that is also in the LDS array
 *   NFC_WAVE_CONFIG(cfgPtr, lds, cc)  fill the NfcConfig `cc` inside a step function (NfcWaveLds::cfg holds its run-time part)
 *   NFC_WAVE_NOINLINE             keeps the step a function of its own
 *   NFC_WAVE_LDS                  address space qualifier of LDS objects
 */
#ifndef NFC_AMD_WAVE_HPP
#define NFC_AMD_WAVE_HPP

#include "nfc_scan_launch.h"

#if NFC_HIST != NFC_HIST_STORED || NFC_RING_STRIDE != 1u
#error "nfc_wave.hpp needs nfc_core.hpp compiled with NFC_RING_STRIDE 1 and NFC_X_OLD_INDEX = nfc_wave_x_old_index"
#endif

#define NFC_WAVE_CARRY 0u   /* work items = jobs: the lane that continues a stream from its own state */
#define NFC_WAVE_WINDOWS 1u /* work items = entries of the run list: speculative windows */
#define NFC_WAVE_FINAL 2u   /* work items = jobs: the lane that regenerates the state of a stream's last window */

/* behind the regions of nfc_core.hpp: the raw samples the tile at hand displaced in the history (one tile is written ahead
 * into a history NFC_HIST deep; the deepest look-back of the path, NFC-V's 472 samples, reaches them), and the clock of
 * the sample before the tile */
#define NFC_WAVE_XOLD (NFC_R_CORR + NFC_CORR_MAX)
#define NFC_WAVE_XOLD_CLOCK (NFC_WAVE_XOLD + NFC_LANES)
#define NFC_WAVE_RING_FLOATS (NFC_WAVE_XOLD_CLOCK + 1u)

/* what the lanes of the wave share: the decoder's state and where the lane of work stands */
struct NfcWaveUni
{
   NfcStreamState s;
   uint32_t consumed; /* samples of the lane's row taken */
   uint32_t stepped;  /* samples stepped one by one (statistics) */
   uint32_t stopped;  /* 1 retired at rest, 2 handed over */
   uint32_t succ;
   uint32_t takeKey;  /* stage for which the tile's running sums have been looked at (nfc_wave_fast): inside the exact range or not; NFC_FK_NONE: not yet */
   uint32_t walked;   /* ... and what was found: 1 = the raw sums of that stage are walked in the step's order (off the grid, or out of the exact range) */
   uint32_t succVerify; /* sample at which the successor at hand publishes (nothing to ask it before): 0 = not looked up yet */
   uint32_t at;       /* sample of the tile at hand */
   /* bulk paths (nfc_wave_fast.hpp) */
   uint32_t key;      /* stage the values in sum / s0 / s1 belong to */
   uint32_t from;     /* first sample of the tile they are valid for */
   uint32_t clock0;   /* clock of the sample before the tile */
   uint32_t gridSince; /* clock from which on every sample has been on the capture grid */
   uint32_t gatedLo, gatedHi; /* the gates as last evaluated, bit j = sample gatedFrom + j of the tile */
   uint32_t gatedFrom;
   uint32_t which;    /* search bank: detectors whose gates were up at sample whichAt (bit per detector, nfc_wave_search_gate) */
   uint32_t whichAt;
   uint32_t maskValid; /* search bank: the detectors whose gates over the tile at hand (NfcWaveLds::gate) still stand (bit per detector) */
   uint32_t aloneLocked; /* search bank: an NFC-F tracker applied on its own found its preamble complete (the sample is the step's) */
   /* what the two ring taps of a sample (nfc_wave_taps) are formed from on demand: the correlators as they stood when the
    * values of the tile were formed (sample `from`): ring position and running sum of the sample before, whether the ring
    * entry one sample back is that sum; locked stages (slot 0): ring period, distance of the first tap, ring base, first
    * clock whose step writes the ring */
   uint32_t tapPos[6];
   float tapAcc[6];
   uint32_t tapPrev;
   uint32_t tapPeriod, tapShift, tapBase, tapWriteFrom;
   float pass[16];    /* hand-over from single lanes to everybody */
};

/* Where the histories of the filtered signal, its deviation and the modulation depth are found beyond the NFC_HIST_F
 * samples the rings hold (NFC-V's look-back of up to 402 samples): the front end's planes of the job, the job's samples,
 * and - before the submission began - the stream's stored history (a carry lane has a copy) */
struct NfcWaveDeep
{
   const float *planes;   /* record of stream position 0 */
   const uint8_t *data;   /* the job's samples, stream position 0 */
   const float *stored;   /* [region][NFC_HIST_STORED] rows of `storedPitch` floats: the history up to the submission's first sample */
   uint64_t storedPitch;
   uint32_t stride;
   uint32_t clockBase;    /* clock of the sample before the submission */
   uint32_t ringEnd;      /* clock of the last sample written to the rings (the tile at hand is written ahead) */
   uint32_t reserved;
};

/* LDS of one wave */
struct NfcWaveLds
{
   float ring[NFC_WAVE_RING_FLOATS]; /* the regions of nfc_core.hpp (NFC_R_*), one stream */
   NfcWaveDeep deep;
   NfcStreamCold cold;
   uint8_t bytes[NFC_STREAM_BYTES];
   uint32_t flags;                   /* NfcStreamCold::usedTech while the lane runs */
   NfcWaveUni u;
   float env[NFC_LANES];             /* envelope / average after each sample of the tile at hand */
   float avg[NFC_LANES];
   uint32_t gate[NFC_LANES];         /* search bank: the detectors' gates per sample of the tile (nfc_wave_search_bits) */
   float sum[6][NFC_LANES];          /* bulk paths: running sum after each sample of the tile, per correlator (the two
                                        differences the detectors look at are formed from it and the ring where they are
                                        used: nfc_wave_s0s1) */
   /* the run-time part of the configuration (enable mask, thresholds), parked where the step functions find it without a
    * trip to memory: [0] enabled, [1] power, [2] low, [3] high threshold, [4..7] correlation, [8..11] minimum, [12..15] maximum depth */
   uint32_t cfg[16];
#ifdef NFC_WAVE_PROFILE
   uint64_t prof[16];
   uint64_t profLast;
   uint32_t profPhase;
#endif
};

/* the frame sink of the launch */
struct NfcWaveSink
{
   uint32_t *words;
   uint32_t *ctl;
   uint32_t capacity;
   uint32_t streamId;
};

/* filtered signal / deviation / modulation depth of the sample of clock `clk` from outside the rings */
template <class Deep>
NFC_DEV float nfc_wave_deep_fetch(const Deep &dc, uint32_t region, uint32_t clk)
{
   const int32_t pos = (int32_t)(clk - dc.clockBase - 1u); /* stream position inside the submission */

   if (pos < 0)
   {
      const uint32_t row = region == NFC_R_FILT ? 1u : (region == NFC_R_MDEV ? 2u : 3u);
      return dc.stored ? dc.stored[(uint64_t)(row * NFC_HIST_STORED + (clk & (NFC_HIST_STORED - 1u))) * dc.storedPitch] : 0.0f;
   }

   const float *p = dc.planes + 4u * (uint64_t)(uint32_t)pos;

   if (region == NFC_R_FILT)
      return p[0];
   if (region == NFC_R_MDEV)
      return p[2];

   /* modulation depth as the front end forms it (NfcTech.cpp:79-83) */
   const float x = NFC_SAMPLE_AT(dc.data, dc.stride, (uint32_t)pos);
   const float env = p[1];
   const float clamped = (x < 0.0f) ? 0.0f : ((env < x) ? env : x);
   return (env - clamped) / env;
}

/* ... of the sample of clock `clk`, wherever it is: in the rings (they hold the NFC_HIST_F samples up to the end of the tile
 * at hand) or beyond them */
NFC_DEV float nfc_wave_f_read(const NFC_WAVE_LDS NfcWaveLds *lds, uint32_t region, uint32_t clk)
{
   if ((uint32_t)(lds->deep.ringEnd - clk) < NFC_HIST_F)
      return lds->ring[region + (clk & NFC_FMASK)];
   return nfc_wave_deep_fetch(lds->deep, region, clk);
}

/* the step machine's reads (NFC_F_DEEP of nfc_core.hpp) */
template <class Mem>
NFC_DEV float nfc_wave_f_deep(const Mem &mem, uint32_t region, uint32_t clk)
{
   if ((uint32_t)(mem.deep->ringEnd - clk) < NFC_HIST_F)
      return mem.ring[region + (clk & NFC_FMASK)];
   return nfc_wave_deep_fetch(*mem.deep, region, clk);
}

NFC_DEV bool nfc_wave_exact_span(uint32_t clock, uint32_t count)
{
   const uint32_t start = clock + 1u + 1024u;
   const uint32_t untilWrap = 0u - start;
   return count != 0 && (start < 2048u || untilWrap < count);
}

/* what a lane works on */
struct NfcWaveItem
{
   uint32_t w;         /* lane slot: index into L.states / L.cold / L.windows / L.works */
   uint32_t mode;
   const uint8_t *data;
   uint32_t count;     /* samples of the row */
   const uint32_t *tiles;
   uint32_t startPos;  /* stream position of the row's first sample */
   uint32_t clockBase; /* clock of the sample before the submission */
   const float *planes; /* record of stream position 0 */
   const NfcScanJob *job;
};

NFC_DEV NfcLaneMem nfc_wave_mem(NFC_WAVE_LDS NfcWaveLds *lds, const NfcWaveSink &sink, const NfcConfig *cfgPtr)
{
   NfcLaneMem mem;
   mem.ring = (NFC_RING_FLOAT *)lds->ring;
   mem.lane = 0;
   mem.exact = false;
   mem.linked = true;
   mem.flags = (uint32_t *)&lds->flags;
   mem.bytes = (uint8_t *)lds->bytes;
   mem.sink = sink.words;
   mem.sinkCursor = sink.ctl;
   mem.sinkDropped = sink.ctl + 1;
   mem.sinkWords = sink.capacity;
   mem.streamId = sink.streamId;
   mem.cold = (NfcStreamCold *)&lds->cold;
   mem.tables = cfgPtr;
   mem.deep = &lds->deep;
   return mem;
}

/* the decoder's edge time after the sample at stream position `last`: the edge-peak tracker (NfcTech.cpp:86-104) walked
 * from the stored point at or before it over the filtered plane, then what the decoder's own copy holds (zeroed by the
 * last carrier frame unless the tracker has moved since: nfc_edge_time). Called by every lane. */
NFC_DEV uint32_t nfc_wave_edge_time(const NfcConfig &c, const NfcScanArgs &A, const NfcWaveItem &it, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t last, uint32_t *trackerTime = nullptr)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t q = last / NFC_SCAN_POINT;
   const NfcScanPoint &pt = A.points[it.job->firstPoint + q];

   uint32_t tracked = (pt.zone & NFC_ZONE_EDGE_KNOWN) ? pt.edgeTime : A.chunkEdge[it.job->firstChunk + (q * NFC_SCAN_POINT) / A.params.chunkSamples];
   float peak = pt.edgePeak;

   /* (16 samples at a time through NfcWaveUni::pass: a rare walk - carrier frames, published and final states) */
   for (uint32_t base = q * NFC_SCAN_POINT; base <= last; base += 16u)
   {
      const uint32_t n = last - base + 1u < 16u ? last - base + 1u : 16u;

      NFC_WAVE_BARRIER();
      if (lane < 16u)
         lds->u.pass[lane] = lane < n ? nfc_abs(it.planes[4u * (uint64_t)(base + lane)]) : 0.0f;
      NFC_WAVE_BARRIER();

      for (uint32_t k = 0; k < n; k++)
      {
         const float rectified = lds->u.pass[k];
         const uint32_t clock = it.clockBase + 1u + base + k;
         const bool high = rectified > c.highThreshold;
         const bool top = high && rectified > peak;
         const bool low = !high && rectified < c.lowThreshold;

         tracked = top ? clock : tracked;
         peak = top ? rectified : (low ? 0.0f : peak);
      }
   }

   NFC_WAVE_BARRIER(); /* (pass[] is free again) */

   const bool emitValid = lds->cold.emitValid != 0;
   const uint32_t emitClock = lds->cold.emitClock;

   if (trackerTime)
      *trackerTime = tracked; /* (the tracker's own time, whatever the last carrier frame has zeroed) */

   return (emitValid && (int32_t)(tracked - emitClock) <= 0) ? 0u : tracked;
}

/* a stream's rings between HBM ([slot][64 lanes], history NFC_HIST_STORED deep) and LDS (history NFC_HIST deep): the
 * stored history holds the last NFC_HIST_STORED samples up to `clock` */
NFC_DEV void nfc_wave_rings_in(NFC_WAVE_LDS NfcWaveLds *lds, const float *src, uint64_t pitch, uint32_t clock, uint32_t corrTotal)
{
   const uint32_t lane = NFC_WAVE_LANE();

   for (uint32_t i = lane; i < NFC_WAVE_RING_FLOATS; i += NFC_LANES)
      lds->ring[i] = 0.0f;

   NFC_WAVE_BARRIER();

   for (uint32_t k = lane; k < NFC_HIST_STORED; k += NFC_LANES)
   {
      const uint32_t t = clock - k; /* sample clock */
      lds->ring[NFC_R_X + (t & NFC_HMASK)] = src[(uint64_t)(t & (NFC_HIST_STORED - 1u)) * pitch];
   }

   /* (the shorter histories: the last NFC_HIST_F samples; what lies further back stays where it is, NfcWaveDeep::stored) */
   for (uint32_t r = 1; r < 4u; r++)
   {
      for (uint32_t k = lane; k < NFC_HIST_F; k += NFC_LANES)
      {
         const uint32_t t = clock - k;
         lds->ring[NFC_R_FILT + (r - 1u) * NFC_HIST_F + (t & NFC_FMASK)] = src[(uint64_t)(r * NFC_HIST_STORED + (t & (NFC_HIST_STORED - 1u))) * pitch];
      }
   }

   for (uint32_t k = lane; k < NFC_PROD + corrTotal; k += NFC_LANES)
      lds->ring[NFC_R_PROD + k] = src[(uint64_t)(4u * NFC_HIST_STORED + k) * pitch];

   NFC_WAVE_BARRIER();
}

NFC_DEV void nfc_wave_rings_out(const NFC_WAVE_LDS NfcWaveLds *lds, float *dst, uint64_t pitch, uint32_t clock, uint32_t corrTotal)
{
   const uint32_t lane = NFC_WAVE_LANE();

   NFC_WAVE_BARRIER();

   for (uint32_t k = lane; k < NFC_HIST_STORED; k += NFC_LANES)
   {
      const uint32_t t = clock - k;
      dst[(uint64_t)(t & (NFC_HIST_STORED - 1u)) * pitch] = lds->ring[NFC_R_X + (t & NFC_HMASK)];
   }

   /* the stored histories are NFC_HIST_STORED deep: what the rings no longer hold is fetched where it is kept (a sample
    * from before the submission is in the stored history already, at the very place: nfc_wave_deep_fetch) */
   for (uint32_t r = 1; r < 4u; r++)
   {
      const uint32_t region = NFC_R_FILT + (r - 1u) * NFC_HIST_F;

      for (uint32_t k = lane; k < NFC_HIST_STORED; k += NFC_LANES)
      {
         const uint32_t t = clock - k;
         const bool before = (int32_t)(t - lds->deep.clockBase - 1u) < 0;

         if (!before || lds->deep.stored != dst)
            dst[(uint64_t)(r * NFC_HIST_STORED + (t & (NFC_HIST_STORED - 1u))) * pitch] = nfc_wave_f_read(lds, region, t);
      }
   }

   for (uint32_t k = lane; k < NFC_PROD + corrTotal; k += NFC_LANES)
      dst[(uint64_t)(4u * NFC_HIST_STORED + k) * pitch] = lds->ring[NFC_R_PROD + k];
}

/* The tile at hand: this lane's sample and the front end's results for it, parked where the step reads them. Returns
 * true when the sample is on the capture grid (nfc_wave_fast.hpp). */
/* this lane's sample of a tile and what the front end made of it (the planes), fetched while the tile before is decoded */
struct NfcWaveFetch
{
   float x, filt, env, mdev, avg;
};

NFC_DEV NfcWaveFetch nfc_wave_fetch(const NfcWaveItem &it, uint32_t consumed, uint32_t stride)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t left = consumed < it.count ? it.count - consumed : 0u;
   NfcWaveFetch f = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

   if (lane < left)
   {
      const uint32_t i = consumed + lane;
      const float *p = it.planes + 4u * (uint64_t)(it.startPos + i);

      f.x = NFC_SAMPLE_AT(it.data, stride, i);
      f.filt = p[0];
      f.env = p[1];
      f.mdev = p[2];
      f.avg = p[3];
   }

   return f;
}

NFC_DEV bool nfc_wave_load_tile(const NfcWaveFetch &f, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t n, uint32_t clock)
{
   const uint32_t lane = NFC_WAVE_LANE();
   bool onGrid = true;

#ifdef NFC_WAVE_DEBUG_FETCH
   NFC_WAVE_DEBUG_FETCH(f, clock);
#endif
   if (lane < n)
   {
      const float x = f.x;
      const float env = f.env;

      /* modulation depth as the front end forms it (NfcTech.cpp:79-83) */
      const float clamped = (x < 0.0f) ? 0.0f : ((env < x) ? env : x);

      const uint32_t slot = (clock + 1u + lane) & NFC_HMASK;
      const uint32_t slotF = (clock + 1u + lane) & NFC_FMASK;

      lds->ring[NFC_WAVE_XOLD + lane] = lds->ring[NFC_R_X + slot]; /* the sample NFC_HIST back */
      lds->ring[NFC_R_X + slot] = x;
      lds->ring[NFC_R_FILT + slotF] = f.filt;
      lds->ring[NFC_R_MDEV + slotF] = f.mdev;
      lds->ring[NFC_R_DEPTH + slotF] = (env - clamped) / env;
      lds->env[lane] = env;
      lds->avg[lane] = f.avg;

      const float scaled = x * 32768.0f;
      onGrid = scaled == __builtin_floorf(scaled) && x >= -1.0f && x <= 1.0f;
   }
   else
      lds->ring[NFC_WAVE_XOLD + lane] = lds->ring[NFC_R_X + ((clock + 1u + lane) & NFC_HMASK)]; /* (not displaced: the same value either way) */

   if (lane == 0)
   {
      lds->ring[NFC_WAVE_XOLD_CLOCK] = __builtin_bit_cast(float, clock);
      lds->deep.ringEnd = clock + n;
   }

   return onGrid;
}

/* v - k*p for the k that brings it below p, for v < p + 64 and p >= 22 (ring periods at the sample rate of the table) */
NFC_DEV uint32_t nfc_wave_wrap3(uint32_t v, uint32_t p)
{
   v -= v >= p ? p : 0u;
   v -= v >= p ? p : 0u;
   v -= v >= p ? p : 0u;
   return v;
}

NFC_DEV uint32_t nfc_wave_wrap1(uint32_t v, uint32_t p)
{
   return v - (v >= p ? p : 0u);
}

/* ring positions after `n` (<= 64) more samples (the incremental form: nfc_bump n times) */
template <class S>
NFC_DEV void nfc_wave_advance(const NfcConfig &c, S &s, uint32_t n)
{
   s.posA[0] = nfc_wave_wrap3(s.posA[0] + n, c.a[0].p1);
   s.posA[1] = nfc_wave_wrap3(s.posA[1] + n, c.a[1].p1);
   s.posA[2] = nfc_wave_wrap3(s.posA[2] + n, c.a[2].p1);
   s.posF[0] = nfc_wave_wrap3(s.posF[0] + n, c.f[1].p1);
   s.posF[1] = nfc_wave_wrap3(s.posF[1] + n, c.f[2].p1);
   s.posV1 = nfc_wave_wrap3(s.posV1 + n, c.v.p1);
   s.posV0 = nfc_wave_wrap3(s.posV0 + n, c.v.p0);
}

/* -DNFC_WAVE_PROFILE: shader cycles per phase of the lane loop, summed over all waves into L.laneStats[12 + phase]
 * (units of 1024 cycles): 0 tile boundary, 1 tile load, 2 bulk values, 3 bulk gates, 4 bulk commit, 5 the step,
 * 6 the search step from values, 7 lane set-up and result, 8 bulk prologue, 9 gates of a locked stage, 10 NFC-B detectors
 * stepped on their own, 11 between a step and the next call */
#ifdef NFC_WAVE_PROFILE
#define NFC_WAVE_TICK(lds, phase)                                  \
   do                                                              \
   {                                                               \
      const uint64_t nowTick = __builtin_readcyclecounter();       \
      if (NFC_WAVE_LANE() == 0)                                    \
      {                                                            \
         (lds)->prof[(lds)->profPhase] += nowTick - (lds)->profLast; \
         (lds)->profLast = nowTick;                                \
         (lds)->profPhase = (phase);                               \
      }                                                            \
   } while (0)
#else
#define NFC_WAVE_TICK(lds, phase) ((void)0)
#endif

/* (the fibre build runs the lanes of a wave one after the other between barriers: every lane has to have read a shared
 * word before one of them goes on to change it; on the GPU the lanes of a wave read together) */
#ifndef NFC_WAVE_DEBUG_POINT
#define NFC_WAVE_DEBUG_POINT(lds, tag) ((void)0)
#endif
#ifndef NFC_WAVE_READ_FENCE
#define NFC_WAVE_READ_FENCE() ((void)0)
#endif

/* statistics of the fibre build (tests/hostsim): samples committed in bulk (0) / stepped (1) per stage */
#ifndef NFC_WAVE_COUNT
#define NFC_WAVE_COUNT(key, which, count) ((void)0)
#endif

#include "nfc_wave_fast.hpp"

/* One sample, by the step machine of nfc_core.hpp: the shared state is taken into registers, stepped, put back.
 * kind: 0 the common step, 1 ring positions by exact modulo (stream start), 2 correlator upkeep only (a window's warm-up).
 * A function of its own: its registers are not the bulk paths' registers. */
NFC_WAVE_NOINLINE void nfc_wave_step(const NfcConfig *cfgPtr, NFC_WAVE_LDS NfcWaveLds *lds, NfcWaveSink sink, uint32_t kind, uint32_t emits, uint32_t edge)
{
   NfcConfig cc;
   NFC_WAVE_CONFIG(cfgPtr, lds, cc);

   const NfcLaneMem mem = nfc_wave_mem(lds, sink, cfgPtr);

   NFC_WAVE_TICK(lds, 12u); /* (profile build: 5 call and configuration, 12 state in, 13 the step machine, 14 state out) */

   NFC_WAVE_UNIFORM_BEGIN
   {
#ifdef NFC_WAVE_STEP_COPY
      NfcStreamState s = *(NfcStreamState *)&lds->u.s;
#else
      NfcStreamState &s = *(NfcStreamState *)&lds->u.s;
#endif
      const uint32_t at = lds->u.at;
      const uint32_t slot = (s.clock + 1u) & NFC_HMASK;
      const uint32_t slotF = (s.clock + 1u) & NFC_FMASK;

      NfcGiven g;
      g.now.x = lds->ring[NFC_R_X + slot];
      g.now.filt = lds->ring[NFC_R_FILT + slotF];
      g.now.mdev = lds->ring[NFC_R_MDEV + slotF];
      g.now.depth = lds->ring[NFC_R_DEPTH + slotF];
      g.env = lds->env[at];
      g.avg = lds->avg[at];

      if (emits)
         s.edgeTime = edge; /* a carrier frame is stamped with it (NfcDecoder.cpp:472-523) */

      NFC_WAVE_TICK(lds, 13u);

      if (kind == 2u)
         nfc_step_upkeep<false, true>(cc, s, mem, g.now.x, &g);
      else if (kind == 1u)
         nfc_step_impl<true, true>(cc, s, mem, g.now.x, &g);
      else
         nfc_step_impl<false, true>(cc, s, mem, g.now.x, &g);

      NFC_WAVE_TICK(lds, 14u);

#ifdef NFC_WAVE_STEP_COPY
      *(NfcStreamState *)&lds->u.s = s;
#endif
      lds->u.maskValid = 0u;
      lds->u.at = at + 1u;
      lds->u.stepped++;
   }
   NFC_WAVE_UNIFORM_END
}

/* One sample of the search bank, from the values the bulk path has formed for the tile (sums and correlations of the
 * six box-sum correlators at this sample): nfc_step_impl / nfc_search_detect (NfcDecoder.cpp:394-418) without walking the
 * correlators again - the decisions are the detectors' own (nfc*_detect_decide). Requires NfcWaveUni::key ==
 * NFC_FK_SEARCH with values valid at this sample; leaves the key at NFC_FK_NONE when they are not valid for the next. */
NFC_WAVE_NOINLINE void nfc_wave_search_step(const NfcConfig *cfgPtr, NFC_WAVE_LDS NfcWaveLds *lds, NfcWaveSink sink, uint32_t emits, uint32_t edge)
{
   NfcConfig c;
   NFC_WAVE_CONFIG(cfgPtr, lds, c);

   const NfcLaneMem mem = nfc_wave_mem(lds, sink, cfgPtr);

   NFC_WAVE_UNIFORM_BEGIN
   {
      /* (in registers for the step: the detectors touch most of the record, and every access in place is an LDS round
       * trip the next one waits for) */
#ifdef NFC_WAVE_SEARCH_IN_PLACE
      NfcStreamState &s = *(NfcStreamState *)&lds->u.s;
#else
      NfcStreamState s = *(NfcStreamState *)&lds->u.s;
#endif
      const uint32_t at = lds->u.at;

      ++s.clock;

      s.posA[0] = nfc_wave_wrap1(s.posA[0] + 1u, c.a[0].p1);
      s.posA[1] = nfc_wave_wrap1(s.posA[1] + 1u, c.a[1].p1);
      s.posA[2] = nfc_wave_wrap1(s.posA[2] + 1u, c.a[2].p1);
      s.posF[0] = nfc_wave_wrap1(s.posF[0] + 1u, c.f[1].p1);
      s.posF[1] = nfc_wave_wrap1(s.posF[1] + 1u, c.f[2].p1);
      s.posV1 = nfc_wave_wrap1(s.posV1 + 1u, c.v.p1);
      s.posV0 = nfc_wave_wrap1(s.posV0 + 1u, c.v.p0);

      const uint32_t slot = s.clock & NFC_FMASK; /* (of the filtered / deviation / depth histories) */

      s.env = lds->env[at];
      s.avg = lds->avg[at];
      s.mdev = lds->ring[NFC_R_MDEV + slot];

      if (emits)
         s.edgeTime = edge;

      /* the detectors whose gates were up when the bulk path looked at this very sample (all of them when it has not:
       * a sample stepped in the wake of another): the others would take their early exits */
      const uint32_t ask = lds->u.whichAt == at ? lds->u.which : 0xFFFFFFFFu;

      nfc_detect_carrier(c, s, mem);

      const bool armed = s.clock >= 1024u && !(s.env < c.powerThreshold);
      uint32_t locked = 0;

      if (armed)
      {
         NfcSearchRegs &r = s.u.search;

         /* first detector that recognises its start of frame wins, later ones skip this sample (their correlators too) */
         if (c.enabled & 1u)
         {
            const float limit = s.env * c.corrThreshold[0];

#define NFC_WAVE_A_RATE(R)                                                                                                                       \
            if (!locked)                                                                                                                         \
            {                                                                                                                                    \
               r.detA[R].acc = lds->sum[R][at];                                                                                                  \
               lds->ring[NFC_R_CORR + c.corrOffset[R] + s.posA[R]] = r.detA[R].acc;                                                              \
               if (((ask >> R) & 1u) && nfca_detect_decide<R>(c, s, mem, nfc_wave_search_num(c, lds, R, at),                                     \
                                         lds->ring[NFC_R_DEPTH + ((s.clock - c.a[R].delay - c.a[R].p8) & NFC_FMASK)], limit, c.minDepth[0]))     \
                  locked = NFC_TECH_A;                                                                                                           \
            }
            NFC_WAVE_A_RATE(0)
            NFC_WAVE_A_RATE(1)
            NFC_WAVE_A_RATE(2)
#undef NFC_WAVE_A_RATE
         }

         if (!locked && (c.enabled & 2u))
         {
            const uint32_t slot0 = (s.clock - c.b[0].delay) & NFC_FMASK, slot1 = (s.clock - c.b[1].delay) & NFC_FMASK;
            const int r0 = ((ask >> 3) & 1u) ? nfcb_detect_decide<0>(c, s, mem, lds->ring[NFC_R_FILT + slot0], lds->ring[NFC_R_DEPTH + slot0]) : 0;

            if (r0 == 1)
               locked = NFC_TECH_B;
            else if (r0 == 0 && ((ask >> 4) & 1u) && nfcb_detect_decide<1>(c, s, mem, lds->ring[NFC_R_FILT + slot1], lds->ring[NFC_R_DEPTH + slot1]) == 1)
               locked = NFC_TECH_B;
         }

         if (!locked && (c.enabled & 4u))
         {
            const float limit = s.env * c.corrThreshold[2];
            const float deep = lds->ring[NFC_R_DEPTH + slot];

            /* (a detector that is not asked may still have been told to reset a record that is clear: the mark it leaves) */
            if (ask != 0xFFFFFFFFu)
            {
               lds->flags |= ((ask >> 8) & 1u) << 16 | ((ask >> 9) & 1u) << 17;

               /* ... or have looked at a record the lane inherited (nfc_wave_gate_f) */
               const uint32_t seen = lds->flags;
               lds->flags = seen | (((ask >> 10) & 1u & ~(seen >> 16)) << 14) | (((ask >> 11) & 1u & ~(seen >> 17)) << 15);
            }

            r.detF[0].acc = lds->sum[3][at];
            lds->ring[NFC_R_CORR + c.corrOffset[3] + s.posF[0]] = r.detF[0].acc;

            float f0, f1;
            nfc_wave_s0s1(c, lds, NFC_FK_SEARCH, 3u, at, f0, f1);

            if (((ask >> 5) & 1u) && nfcf_detect_decide<1>(c, s, mem, f0, f0 - f1, deep, limit))
               locked = NFC_TECH_F;
            else
            {
               r.detF[1].acc = lds->sum[4][at];
               lds->ring[NFC_R_CORR + c.corrOffset[4] + s.posF[1]] = r.detF[1].acc;

               nfc_wave_s0s1(c, lds, NFC_FK_SEARCH, 4u, at, f0, f1);

               if (((ask >> 6) & 1u) && nfcf_detect_decide<2>(c, s, mem, f0, f0 - f1, deep, limit))
                  locked = NFC_TECH_F;
            }
         }

         if (!locked && (c.enabled & 8u))
         {
            r.detV.acc = lds->sum[5][at];
            lds->ring[NFC_R_CORR + c.corrOffset[5] + s.posV1] = r.detV.acc;

            float v0, v1;
            nfc_wave_s0s1(c, lds, NFC_FK_SEARCH, 5u, at, v0, v1);

            if (((ask >> 7) & 1u) && nfcv_detect_decide(c, s, mem, v0, lds->ring[NFC_R_X + ((s.clock - c.v.delay) & NFC_HMASK)]))
               locked = NFC_TECH_V;
         }

         if (!locked)
         {
            if (s.bankClock != s.clock - 1u)
               mem.cold->bankRun = s.clock;
            s.bankClock = s.clock;
         }
      }

      if (locked)
         nfc_enter_lock(s, mem, locked);

      /* the values stay good for the next sample unless a detector locked (some correlators then skipped this sample)
       * or the bank did not step at all */
      if (locked || !armed)
         lds->u.key = NFC_FK_NONE;

      /* the gates of the detectors that were asked no longer stand (NFC-F: the marks with them) */
      lds->u.maskValid &= ~(ask & 0xFFu);

#ifndef NFC_WAVE_SEARCH_IN_PLACE
      *(NfcStreamState *)&lds->u.s = s;
#endif
      lds->u.at = at + 1u;
      lds->u.stepped++;
   }
   NFC_WAVE_UNIFORM_END
}

/* One tile: the next n samples of the lane's row (stream position pos on). allowFast: take the bulk paths (the fibre
 * build runs every tile a second time without them and compares: tests/hostsim/emu_wave.cpp). */
NFC_DEV void nfc_wave_tile(const NfcConfig *cfgPtr, const NfcConfig &cc, const NfcScanArgs &A, const NfcWaveItem &it, NFC_WAVE_LDS NfcWaveLds *lds,
                           const NfcWaveSink &sink, uint32_t n, uint32_t pos, bool carry, uint32_t warmFront, uint32_t warm, const NfcWaveFetch &fetched, bool allowFast)
{
   const uint32_t consumed = NFC_WAVE_UNIFORM_U32(lds->u.consumed);
   const uint32_t clock = NFC_WAVE_UNIFORM_U32(lds->u.s.clock);

   NFC_WAVE_TICK(lds, 1u);
   NFC_WAVE_BARRIER();
   const bool onGrid = nfc_wave_load_tile(fetched, lds, n, clock);
   const bool allOnGrid = NFC_WAVE_BALLOT(!onGrid) == 0ull;
   NFC_WAVE_BARRIER();
   NFC_WAVE_DEBUG_POINT(lds, 4u);

   /* Ring positions by exact modulo (nfc_core.hpp, nfc_exact_zone): around the wrap of the 32-bit clock on every sample;
    * at the start of a stream only to put the positions of a fresh state right - the first step of the tile does that,
    * from there on counting along gives the same positions. */
   const bool exactSpan = carry && nfc_wave_exact_span(clock, n);
   const bool streamStart = (uint32_t)(clock + 1u + 1024u) < 2048u;

   NFC_WAVE_COUNT(43u, 0u, 1u); /* tiles */

   /* per tile: the values of the bulk paths belong to the tile; is the tile on the grid? */
   NFC_WAVE_UNIFORM_BEGIN
   {
      lds->u.key = NFC_FK_NONE;
      lds->u.from = 0;
      lds->u.clock0 = clock;
      lds->u.at = 0;
      lds->u.gatedLo = 0;
      lds->u.gatedHi = 0;
      lds->u.gatedFrom = 0;
      lds->u.which = 0xFFFFFFFFu;
      lds->u.whichAt = 0xFFFFFFFFu;
      lds->u.maskValid = 0u;
      lds->u.takeKey = NFC_FK_NONE;
      lds->u.walked = 0u;
      if (!allOnGrid)
         lds->u.gridSince = clock + n;
   }
   NFC_WAVE_UNIFORM_END

   if (consumed < warmFront)
   {
      /* history only (nfc_step_front) */
      NFC_WAVE_UNIFORM_BEGIN
      {
         lds->u.s.clock = clock + n;
         nfc_wave_advance(cc, lds->u.s, n);
         lds->u.s.env = lds->env[n - 1u];
         lds->u.s.avg = lds->avg[n - 1u];
         lds->u.s.mdev = lds->ring[NFC_R_MDEV + ((clock + n) & NFC_FMASK)];
      }
      NFC_WAVE_UNIFORM_END
      return;
   }

   const bool upkeep = consumed < warm;
   bool again = false; /* the sample at hand was gated when the gates were last evaluated */

   for (;;)
   {
      uint32_t at = NFC_WAVE_UNIFORM_U32(lds->u.at);
      NFC_WAVE_READ_FENCE();

      if (at >= n)
         break;

      const bool exact = exactSpan && (!streamStart || at == 0u || !allowFast);

      /* Samples from `at` on that change nothing but sums and rings: committed in bulk. Gated samples tend to come in
       * runs (a detector following a pulse): one that was gated at the last evaluation is stepped without asking
       * again - a step is right on any sample, gated or not. */
      if (!exact && allowFast && !again)
      {
         const bool went = nfc_wave_fast(cc, lds, n, upkeep);
         NFC_WAVE_DEBUG_POINT(lds, 1u);
         if (went)
            continue;

         at = NFC_WAVE_UNIFORM_U32(lds->u.at); /* (it may have committed a run before the sample to step) */
         NFC_WAVE_READ_FENCE();
      }

      /* carrier frame due on this sample (NfcDecoder.cpp:472-523)? it is stamped with the decoder's edge time */
      const float avgAt = lds->avg[at];
      const bool emits = !upkeep && lds->u.s.lockTech == 0 &&
                         ((avgAt > cc.highThreshold) ? !lds->u.s.carrierOn : ((avgAt < cc.lowThreshold) && !lds->u.s.carrierOff));
      uint32_t edge = 0;

      if (emits)
         edge = nfc_wave_edge_time(cc, A, it, lds, pos + at);

      NFC_WAVE_COUNT(nfc_wave_stage(lds->u.s, upkeep), 1u, 1u);
      if (again)
         NFC_WAVE_COUNT(45u, 0u, 1u); /* stepped in the wake of another */

      /* the search bank from the values the bulk path holds for this sample, or the step machine itself */
      const bool fromValues = allowFast && !exact && !upkeep && lds->u.s.lockTech == 0 && lds->u.s.unlock == 0 &&
                              NFC_WAVE_UNIFORM_U32(lds->u.key) == NFC_FK_SEARCH && NFC_WAVE_UNIFORM_U32(lds->u.from) <= at;

      NFC_WAVE_TICK(lds, fromValues ? 6u : 5u);
      NFC_WAVE_READ_FENCE(); /* (everything above has been read from the state as it stood) */

      if (fromValues)
         nfc_wave_search_step(cfgPtr, lds, sink, emits ? 1u : 0u, edge);
      else
         nfc_wave_step(cfgPtr, lds, sink, upkeep ? 2u : (exact ? 1u : 0u), emits ? 1u : 0u, edge);

      NFC_WAVE_TICK(lds, 11u);
      NFC_WAVE_DEBUG_POINT(lds, fromValues ? 2u : 3u);

      {
         /* (the search bank keeps its gates per detector: asking again is cheap there, and tells which detectors to ask) */
         const uint32_t keyNow = NFC_WAVE_UNIFORM_U32(lds->u.key);

         again = false;

         if (allowFast && !exact && keyNow != NFC_FK_SEARCH && keyNow != NFC_FK_NONE)
         {
            const uint64_t gated = ((uint64_t)NFC_WAVE_UNIFORM_U32(lds->u.gatedHi) << 32) | NFC_WAVE_UNIFORM_U32(lds->u.gatedLo);
            const uint32_t next = at + 1u - NFC_WAVE_UNIFORM_U32(lds->u.gatedFrom);
            /* (only while the decoder stays in the stage the gates were evaluated for) */
            again = next < 64u && ((gated >> next) & 1ull) != 0ull && keyNow == NFC_WAVE_UNIFORM_U32(nfc_wave_stage(NFC_WAVE_STATE(lds), upkeep));
         }
      }
   }
}

/* One lane of work. `lds`: this wave's LDS. Called by all 64 lanes. */
NFC_DEV void nfc_wave_run(const NfcConfig *cfgPtr, const NfcConfig &cc, const NfcLaunch &L, const NfcScanArgs &A, uint32_t mode, uint32_t item,
                          NFC_WAVE_LDS NfcWaveLds *lds)
{
   const uint32_t lane = NFC_WAVE_LANE();

   NfcWaveItem it;
   it.mode = mode;

   if (mode == NFC_WAVE_WINDOWS)
   {
      if (item >= *A.runCount)
         return;
      it.w = A.runList[item];
   }
   else if (mode == NFC_WAVE_FINAL)
      it.w = A.finalLaneSlot + item;
   else
      it.w = item;

   {
      const NfcWork work = L.works[it.w];
      it.data = work.data;
      it.count = work.count;
      it.tiles = work.tiles;
   }

   if (it.count == 0)
      return;

   const bool carry = mode == NFC_WAVE_CARRY;
   const uint32_t stride = L.uniformStride;

   NfcWindow *me = L.windows + it.w;
   it.job = L.jobs + me->job;
   it.startPos = me->start;
   it.clockBase = A.states[it.job->slot].clock;
   it.planes = A.planes + 4u * (uint64_t)it.job->firstTile * NFC_SCAN_TILE;

   const uint32_t verifyPos = me->verify;
   const uint32_t jobWindows = it.job->windows;
   const uint32_t succEnd = it.job->firstWindow + jobWindows;
   const uint32_t activate = me->activate;
   const uint32_t warmFront = carry ? 0u : NFC_WINDOW_WARM_FRONT;
   const uint32_t warm = carry ? 0u : NFC_WINDOW_WARM_FRONT + NFC_WINDOW_WARM_CORR;

   NfcWaveSink sink;
   sink.words = L.sink;
   sink.ctl = L.sinkCtl;
   sink.capacity = L.sinkWords;
   sink.streamId = it.w;

   const uint32_t startClock = L.states[it.w].clock;

   /* LDS: the decoder's state, the stream's rings (carry lanes: from the lane's copy of the stream's storage), protocol
    * state, frame bytes */
   for (uint32_t i = lane; i < sizeof(NfcStreamState) / 4u; i += NFC_LANES)
      ((NFC_WAVE_LDS uint32_t *)&lds->u.s)[i] = ((const uint32_t *)(L.states + it.w))[i];

   const uint64_t pitch = NFC_LANES;
   float *laneRings = L.rings + (uint64_t)(it.w / NFC_LANES) * L.ringBlockFloats + (it.w % NFC_LANES);

   if (carry)
      nfc_wave_rings_in(lds, laneRings, pitch, startClock, cc.corrTotal);
   else
   {
      for (uint32_t i = lane; i < NFC_WAVE_RING_FLOATS; i += NFC_LANES)
         lds->ring[i] = 0.0f;
   }

   for (uint32_t i = lane; i < sizeof(NfcStreamCold) / 4u; i += NFC_LANES)
      ((NFC_WAVE_LDS uint32_t *)&lds->cold)[i] = ((const uint32_t *)(L.cold + it.w))[i];

   for (uint32_t i = lane; i < NFC_STREAM_BYTES / 4u; i += NFC_LANES)
      ((NFC_WAVE_LDS uint32_t *)lds->bytes)[i] = carry ? ((const uint32_t *)(L.bytes + (uint64_t)it.w * NFC_STREAM_BYTES))[i] : 0u;

#ifdef NFC_WAVE_PROFILE
   if (lane == 0)
   {
      for (int i = 0; i < 16; i++)
         lds->prof[i] = 0;
      lds->profLast = __builtin_readcyclecounter();
      lds->profPhase = 7u;
   }
#endif

   if (lane == 0)
   {
      lds->deep.planes = it.planes;
      lds->deep.data = it.job->data;
      lds->deep.stored = carry ? laneRings : nullptr; /* (only a lane that begins with the submission looks back beyond it) */
      lds->deep.storedPitch = pitch;
      lds->deep.stride = stride;
      lds->deep.clockBase = it.clockBase;
      lds->deep.ringEnd = startClock;
      lds->deep.reserved = 0;

      lds->cfg[0] = cc.enabled;
      lds->cfg[1] = nfc_bits(cc.powerThreshold);
      lds->cfg[2] = nfc_bits(cc.lowThreshold);
      lds->cfg[3] = nfc_bits(cc.highThreshold);
      for (int t = 0; t < 4; t++)
      {
         lds->cfg[4 + t] = nfc_bits(cc.corrThreshold[t]);
         lds->cfg[8 + t] = nfc_bits(cc.minDepth[t]);
         lds->cfg[12 + t] = nfc_bits(cc.maxDepth[t]);
      }
   }

   NFC_WAVE_BARRIER();

   NFC_WAVE_UNIFORM_BEGIN
   {
      lds->flags = 0u;
      lds->u.consumed = 0;
      lds->u.stepped = 0;
      lds->u.stopped = 0;
      lds->u.at = 0;
      lds->u.succVerify = 0u;
      lds->u.succ = carry ? it.job->firstWindow : it.w + 1u;
      if (mode == NFC_WAVE_FINAL || (mode == NFC_WAVE_WINDOWS && (it.w < it.job->firstWindow || it.w >= succEnd)))
         lds->u.succ = succEnd; /* runs on its own */

      /* the front-end recurrences are not walked here (planes): what of them a lane's records are compared by is kept in
       * one form by every lane (nfc_lane_digest); a lane that reaches the end of the submission leaves the scanned state */
      lds->u.s.n1 = 0.0f;
      lds->u.s.edgePeak = 0.0f;
      lds->u.s.pulseFilter = 0u;

      /* samples known to be on the capture grid (nfc_wave_fast.hpp): the job's own are (the scan has looked at every one;
       * what it does not check, |x| <= 1, is checked per tile); what a carry lane finds in the stream's rings is not known */
      lds->u.key = NFC_FK_NONE;
      lds->u.from = 0;
      lds->u.clock0 = startClock;
      lds->u.gridSince = carry ? startClock : startClock - 4096u;
   }
   NFC_WAVE_UNIFORM_END

   const NfcLaneMem mem = nfc_wave_mem(lds, sink, cfgPtr);

   NfcWaveFetch fetched = nfc_wave_fetch(it, 0u, stride);

   /* What a tile boundary needs to know, kept in registers from one boundary to the next (round 5: each used to be a word of the
    * shared record, an LDS round trip apiece on every boundary - 9 % of the wave's cycles for a boundary at which, nearly
    * always, nothing happens): where the lane stands, the retire / dark flags of the 64 tiles at hand, the sample from which
    * on the successor is to be asked. */
   uint32_t consumedNow = 0;
   uint64_t retireMask = 0ull, darkMask = 0ull;
   uint32_t askFrom = 0u; /* (0: not looked up yet) */

   for (;;)
   {
      const uint32_t consumed = consumedNow;

      if (consumed >= it.count)
         break;

      const uint32_t pos = it.startPos + consumed;

      /* ---- tile boundary: publish, retire, hand over (nfc_window_body) ---- */
      NFC_WAVE_TICK(lds, 0u);
      const bool past = consumed >= warm && consumed > 0;

      /* (the flag words of the next 64 tiles, fetched together: a load per tile boundary is a memory latency per tile) */
      {
         const uint32_t tile = consumed / NFC_SCAN_TILE;

         if ((tile % NFC_LANES) == 0u || consumed == 0u)
         {
            const uint32_t first = tile / NFC_LANES * NFC_LANES;
            const uint32_t tilesOfRow = (it.count + NFC_SCAN_TILE - 1u) / NFC_SCAN_TILE;
            const uint32_t word = (jobWindows != 0u && first + lane < tilesOfRow) ? it.tiles[first + lane] : 0u;
            const uint64_t retire = NFC_WAVE_BALLOT((word & NFC_TILE_RETIRE_OK) != 0u);
            /* (tiles of the row that exist, whatever the stream has in the way of windows) */
            const uint32_t flagsHere = first + lane < tilesOfRow ? it.tiles[first + lane] : 0u;
            const uint64_t dark = NFC_WAVE_BALLOT((flagsHere & NFC_TILE_DARK) != 0u);

            retireMask = retire;
            darkMask = dark;
         }
      }

      /* (a stream without windows - NfcScanParams::soloSamples - has nobody to take over from a lane that retires) */
      const uint32_t tileBit = (consumed / NFC_SCAN_TILE) % NFC_LANES;
      const bool mayRetire = past && jobWindows != 0u && ((retireMask >> tileBit) & 1ull) != 0ull;
      const bool publishes = pos == verifyPos;
      uint32_t edgeNow = 0;
      uint32_t stoppedNow = 0u;

      if (publishes && pos > 0)
         edgeNow = nfc_wave_edge_time(cc, A, it, lds, pos - 1u); /* a published state carries the decoder's edge time */

      /* (nothing to publish, no leave to retire, the successor not to be asked yet: nothing to look at) */
      if (publishes || mayRetire || (past && pos >= askFrom))
      {
      uint32_t askNext = askFrom;

      NFC_WAVE_UNIFORM_BEGIN
      {
         NfcStreamState &s = *(NfcStreamState *)&lds->u.s;

         if (publishes && pos > 0)
            s.edgeTime = edgeNow;

         if (publishes)
            nfc_lane_publish(*me, s, *mem.cold);

         if (mayRetire && nfc_quiescent(s) && s.bankClock == s.clock && (uint32_t)(s.clock - mem.cold->bankRun) >= NFC_WINDOW_SETTLE)
            lds->u.stopped = 1;

         /* (the successor is only asked once the lane has reached the sample it publishes at: before that
          * nfc_lane_handover would find nothing to do, at the price of a trip to memory per tile) */
         if (!lds->u.stopped && past && pos >= lds->u.succVerify)
         {
            uint32_t succ = lds->u.succ;
            if (nfc_lane_handover(L.windows, *me, succ, succEnd, pos, s, *mem.cold))
               lds->u.stopped = 2;
            lds->u.succ = succ;
            lds->u.succVerify = succ < succEnd ? L.windows[succ].verify : 0xFFFFFFFFu;
         }

         stoppedNow = lds->u.stopped;
         askNext = lds->u.succVerify;
         NFC_WAVE_UNIFORM_LEAVE(lds->u.pass[0], __builtin_bit_cast(float, stoppedNow));
         NFC_WAVE_UNIFORM_LEAVE(lds->u.pass[1], __builtin_bit_cast(float, askNext));
      }
      NFC_WAVE_UNIFORM_END

      NFC_WAVE_UNIFORM_TAKE_BITS(lds->u.pass[0], stoppedNow);
      NFC_WAVE_UNIFORM_TAKE_BITS(lds->u.pass[1], askNext);
      askFrom = askNext;
      }

      if (stoppedNow)
         break;

      /* ---- a searching lane in front of a run of dark tiles ----
       * In a dark tile (nfc_tile_flags: every sample below the power threshold, no carrier event, on the capture grid) a
       * searching decoder runs its front end and nothing else: the detector bank is not stepped (NfcDecoder.cpp:394-418: not
       * armed), sums and correlation rings stand as they are - stale, exactly like the reference's -, the ring positions count
       * on with the clock. The front end's results are the planes', so the lane goes straight to the last eight dark tiles it
       * knows of (the flags of 64 tiles at a time): their 512 samples put back every history a detector can look into once
       * the envelope is above the threshold again (NFC_HIST; the shorter histories hold 256), the clock and the positions
       * are moved by what was left out. Not across the sample the lane publishes at. (S1's captures begin with the carrier
       * off: a fifth of what the lanes of a dense stream walk is such tiles.) */
      if (past && ((darkMask >> tileBit) & 1ull) != 0ull && NFC_WAVE_STATE(lds).lockTech == 0u && NFC_WAVE_STATE(lds).unlock == 0u)
      {
         const uint64_t from = darkMask >> tileBit;
         const uint32_t run = (~from) ? (uint32_t)__builtin_ctzll(~from) : 64u; /* dark tiles in a row from this one (the group's own: bits beyond it are zero) */
         const uint32_t keep = NFC_HIST / NFC_SCAN_TILE;

         if (run > keep)
         {
            uint32_t tilesLeftOut = run - keep;

            if (verifyPos > pos && verifyPos - pos < tilesLeftOut * NFC_SCAN_TILE)
               tilesLeftOut = (verifyPos - pos) / NFC_SCAN_TILE;
            while (tilesLeftOut && consumed + tilesLeftOut * NFC_SCAN_TILE + NFC_SCAN_TILE > it.count)
               tilesLeftOut--;

            /* No carrier frame may fall due on the way (NfcDecoder.cpp:472-523). The scan's carrier zone - the zone the average
             * was last seen in - does not change inside dark tiles (a change is NFC_TILE_CARRIER), and a decoder whose carrier
             * state agrees with it has nothing to emit whatever the average does between the thresholds. One that has been
             * locked while the average crossed has not emitted yet, and does on its first sample here: it is left to walk. */
            if (tilesLeftOut)
            {
               const uint32_t zone = A.points[it.job->firstPoint + (pos + NFC_SCAN_POINT - 1u) / NFC_SCAN_POINT].zone & NFC_ZONE_MASK;
               const bool agrees = zone == 1u ? NFC_WAVE_STATE(lds).carrierOn != 0u : (zone == 2u ? NFC_WAVE_STATE(lds).carrierOff != 0u : true);

               if (!agrees)
                  tilesLeftOut = 0u;
            }

            if (tilesLeftOut)
            {
               const uint32_t samples = tilesLeftOut * NFC_SCAN_TILE;

               NFC_WAVE_READ_FENCE();
               NFC_WAVE_UNIFORM_BEGIN
               {
                  NFC_WAVE_LDS NfcStreamState &w = lds->u.s;
                  w.clock += samples;
                  w.posA[0] = (w.posA[0] + samples) % cc.a[0].p1;
                  w.posA[1] = (w.posA[1] + samples) % cc.a[1].p1;
                  w.posA[2] = (w.posA[2] + samples) % cc.a[2].p1;
                  w.posF[0] = (w.posF[0] + samples) % cc.f[1].p1;
                  w.posF[1] = (w.posF[1] + samples) % cc.f[2].p1;
                  w.posV1 = (w.posV1 + samples) % cc.v.p1;
                  w.posV0 = (w.posV0 + samples) % cc.v.p0;
                  lds->u.consumed = consumed + samples;
               }
               NFC_WAVE_UNIFORM_END

               consumedNow = consumed + samples;
               fetched = nfc_wave_fetch(it, consumed + samples, stride);
               continue; /* (the boundary of the tile it lands on is looked at like any other) */
            }
         }
      }

      /* ---- the tile ---- */
      const uint32_t left = it.count - consumed;
      const uint32_t n = left < NFC_LANES ? left : NFC_LANES;

      /* (the next tile's samples are on their way while this one is decoded) */
      const NfcWaveFetch ahead = nfc_wave_fetch(it, consumed + n, stride);

#ifdef NFC_WAVE_TILE_HOOK
      NFC_WAVE_TILE_HOOK(cfgPtr, cc, A, it, lds, sink, n, pos, carry, warmFront, warm, fetched);
#else
      nfc_wave_tile(cfgPtr, cc, A, it, lds, sink, n, pos, carry, warmFront, warm, fetched, true);
#endif

      fetched = ahead;
      consumedNow = consumed + n;

      NFC_WAVE_UNIFORM_BEGIN
      {
         lds->u.consumed = consumed + n;
      }
      NFC_WAVE_UNIFORM_END
   }

   /* ---- the lane's result ---- */
   NFC_WAVE_TICK(lds, 7u);
   const uint32_t consumed = consumedNow;
   const uint32_t stopped = NFC_WAVE_UNIFORM_U32(lds->u.stopped);
   const bool ranOut = consumed >= it.count;
   const bool atEnd = it.startPos + consumed >= it.job->count;
   const bool closing = activate >= it.startPos + it.count;

   uint32_t edgeEnd = 0, trackedEnd = me->tracked; /* (a lane that took no sample: what it started with) */
   if (!atEnd && it.startPos + consumed > 0)
      edgeEnd = nfc_wave_edge_time(cc, A, it, lds, it.startPos + consumed - 1u, &trackedEnd);

   NFC_WAVE_UNIFORM_BEGIN
   {
      NfcStreamState &s = *(NfcStreamState *)&lds->u.s;

      if (atEnd)
      {
         /* the front end where the submission ends, as the scan left it */
         const uint32_t lastChunk = it.job->firstChunk + it.job->chunks - 1u;
         const NfcScanPoint &p = A.seams[lastChunk].end;
         const uint32_t tracked = (p.zone & NFC_ZONE_EDGE_KNOWN) ? p.edgeTime : A.chunkEdge[lastChunk];

         s.env = p.env;
         s.n1 = p.n1;
         s.mdev = p.mdev;
         s.avg = p.avg;
         s.edgePeak = p.edgePeak;
         s.pulseFilter = p.pulseFilter;
         s.edgeTime = (mem.cold->emitValid && (int32_t)(tracked - mem.cold->emitClock) <= 0) ? 0u : tracked;
         lds->cold.trackedEnd = tracked;
      }
      else
      {
         if (it.startPos + consumed > 0)
            s.edgeTime = edgeEnd;
         lds->cold.trackedEnd = trackedEnd;
      }

      lds->cold.usedTech = lds->flags;
   }
   NFC_WAVE_UNIFORM_END

   NFC_WAVE_BARRIER();

   /* 2 handed over, 1 stopped at rest - or out of samples in a state the closing window can take over from -, 0 ran to
    * the end of the submission (nfc_window_body) */
   const uint32_t how = stopped == 2 ? 2u : ((!ranOut || (!closing && nfc_lane_comparable(*(const NfcStreamState *)&lds->u.s, *mem.cold))) ? 1u : 0u);
   const uint32_t endClock = lds->u.s.clock;

   for (uint32_t i = lane; i < sizeof(NfcStreamCold) / 4u; i += NFC_LANES)
      ((uint32_t *)(L.cold + it.w))[i] = ((const NFC_WAVE_LDS uint32_t *)&lds->cold)[i];

   for (uint32_t i = lane; i < sizeof(NfcStreamState) / 4u; i += NFC_LANES)
      ((uint32_t *)(L.states + it.w))[i] = ((const NFC_WAVE_LDS uint32_t *)&lds->u.s)[i];

   /* rings and frame bytes: a carry or final lane owns storage; a window that ran to the end of the submission with
    * nobody to take over may be the stream's last lane and leaves a copy in the save area (NfcScanArgs::saveRings) */
   uint32_t saved = 0;

   if (mode != NFC_WAVE_WINDOWS)
   {
      nfc_wave_rings_out(lds, laneRings, pitch, endClock, cc.corrTotal);

      for (uint32_t i = lane; i < NFC_STREAM_BYTES / 4u; i += NFC_LANES)
         ((uint32_t *)(L.bytes + (uint64_t)it.w * NFC_STREAM_BYTES))[i] = ((const NFC_WAVE_LDS uint32_t *)lds->bytes)[i];
   }
   else if (how == 0u && !closing)
   {
      NFC_WAVE_BARRIER();
      NFC_WAVE_UNIFORM_BEGIN
      {
         lds->u.at = NFC_ATOMIC_ADD(A.saveNext, 1u);
      }
      NFC_WAVE_UNIFORM_END

      const uint32_t slot = NFC_WAVE_UNIFORM_U32(lds->u.at);

      if (slot < A.saveRoom)
      {
         const uint32_t rows = L.ringBlockFloats / NFC_LANES;

         nfc_wave_rings_out(lds, A.saveRings + (uint64_t)slot * rows, 1u, endClock, cc.corrTotal);

         for (uint32_t i = lane; i < NFC_STREAM_BYTES / 4u; i += NFC_LANES)
            ((uint32_t *)(A.saveBytes + (uint64_t)slot * NFC_STREAM_BYTES))[i] = ((const NFC_WAVE_LDS uint32_t *)lds->bytes)[i];

         saved = slot + 1u;
      }
   }

   if (lane == 0)
   {
      me->stop = it.startPos + consumed;
      me->retired = how;
      if (mode == NFC_WAVE_WINDOWS)
         me->saved = saved;

#ifdef NFC_WAVE_WHY
      /* debug build: lanes that stepped most of their samples */
      if (lds->u.stepped > 20000u)
         printf("[wave why] slot %u mode %u start %u count %u consumed %u stepped %u lockTech %x stage %u gridSince %u clock %u startClock %u accA %g %g %g accF %g %g accV %g lockacc %g\n", it.w, mode,
                it.startPos, it.count, consumed, lds->u.stepped, lds->u.s.lockTech, nfc_wave_stage(*(const NfcStreamState *)&lds->u.s, false), lds->u.gridSince, lds->u.s.clock, startClock,
                (double)lds->u.s.u.search.detA[0].acc, (double)lds->u.s.u.search.detA[1].acc, (double)lds->u.s.u.search.detA[2].acc, (double)lds->u.s.u.search.detF[0].acc,
                (double)lds->u.s.u.search.detF[1].acc, (double)lds->u.s.u.search.detV.acc, (double)lds->u.s.u.decode.lock.acc);
#endif
      const uint32_t tilesStepped = (lds->u.stepped + NFC_LANES - 1u) / NFC_LANES;
      NFC_WAVE_STAT_ADD(L.laneStats, tilesStepped);
      NFC_WAVE_STAT_MAX(L.laneStats + 1, tilesStepped);
      NFC_WAVE_STAT_ADD(L.laneStats + 2, 1u);
      NFC_WAVE_STAT_ADD(L.laneStats + 6, (consumed + NFC_SCAN_TILE - 1u) / NFC_SCAN_TILE); /* tiles the lane took (warm-up and jumped dark tiles included) */

#ifdef NFC_WAVE_PROFILE
      NFC_WAVE_TICK(lds, 7u);
      for (int i = 0; i < 16; i++)
         NFC_WAVE_STAT_ADD(L.laneStats + 12 + i, (uint32_t)(lds->prof[i] >> 10));
#endif
   }
}

#endif
