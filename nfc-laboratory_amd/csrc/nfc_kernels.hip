/*
 * nfc_kernels.hip — gfx950 kernels of the demodulation path.
 *
 *   nfc_demod_kernel   one wavefront = one stream block (64 capture streams, one lane each).
 *                      Per 64-sample tile: the wave loads 64 rows (one per stream) of IQ/magnitude with
 *                      fully coalesced 512 B / 256 B row reads, turns IQ into magnitude on the fly
 *                      (the reference's RadioDeviceTask.cpp:626-642 scalar formula, no contraction) and parks
 *                      the tile transposed in LDS (pitch 65 -> conflict-free column reads); then every lane
 *                      walks its own row through nfc_step(). History rings live in HBM as [slot][64 lanes]
 *                      so a clock-aligned wave touches one 256 B row per ring access.
 *   nfc_init_kernel    (re)initialise stream slots: decoder state + rings.
 *
 * Bound: HBM. Algorithmic traffic is 8 B (IQ) or 4 B (magnitude) read per sample plus frame bytes written;
 * MFMA is not applicable (1-D sliding correlation + sequential state machine, no GEMM shape).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NFC_DEV __device__ __forceinline__
#define NFC_ATOMIC_ADD(ptr, value) atomicAdd((ptr), (value))
#define NFC_ANY(predicate) (__any(predicate) != 0)

#include "nfc_core.hpp"

/* magnitude of one IQ sample, the reference's scalar formula (RadioDeviceTask.cpp:626-642): products and sum rounded
 * separately (no contraction), correctly rounded square root */
__device__ __forceinline__ float nfc_iq_magnitude(float i, float q)
{
   return __builtin_sqrtf(__fadd_rn(__fmul_rn(i, i), __fmul_rn(q, q)));
}

__device__ __forceinline__ float nfc_sample_at(const uint8_t *data, uint32_t stride, uint32_t index)
{
   if (stride == 2)
   {
      const float2 iq = reinterpret_cast<const float2 *>(data)[index];
      return nfc_iq_magnitude(iq.x, iq.y);
   }
   return reinterpret_cast<const float *>(data)[index];
}

#define NFC_SAMPLE_AT(data, stride, index) nfc_sample_at((data), (stride), (index))
#define NFC_FENCE() __threadfence()
/* a window record written by the 64 lanes of the wave that builds a stream's windows (nfc_windows_kernel; every lane is there with
 * the same values): lane d + 64 k writes word d + 64 k - the record's first four words are the ones that are not zero */
#define NFC_WINDOW_STORE(w, job_, start_, activate_, verify_)                                                                   \
   do                                                                                                                           \
   {                                                                                                                            \
      static_assert(offsetof(NfcWindow, job) == 0 && offsetof(NfcWindow, start) == 4 && offsetof(NfcWindow, activate) == 8 &&  \
                       offsetof(NfcWindow, verify) == 12 && sizeof(NfcWindow) % 4 == 0,                                         \
                    "NFC_WINDOW_STORE writes the first four words of NfcWindow by position");                                   \
      uint32_t *words_ = (uint32_t *)&(w);                                                                                      \
      for (uint32_t d_ = threadIdx.x; d_ < sizeof(NfcWindow) / 4u; d_ += NFC_LANES)                                             \
         words_[d_] = d_ == 0u ? (uint32_t)(job_) : (d_ == 1u ? (uint32_t)(start_) : (d_ == 2u ? (uint32_t)(activate_) : (d_ == 3u ? (uint32_t)(verify_) : 0u))); \
   } while (0)
#include "nfc_scan.hpp"
#include "nfc_launch.h"
#include "nfc_scan_launch.h"

/* sample-rate-derived constants of the most common configuration as literals (generated at build time) */
#define NFC_FIXED_FN __device__ __forceinline__
#include "nfc_config_fixed.inc"

/* constant address space: uniform loads through it are always scalar (s_load), whatever else the kernel writes */
typedef __attribute__((address_space(4))) NfcConfig NfcConfigConst;

/* samples per input tile (<= 64: one row of the tile is fetched by one wave-wide load) */
#ifndef NFC_TILE
#define NFC_TILE 64
#endif
#define TILE NFC_TILE
#define TILE_PITCH (NFC_TILE + 1)

#ifndef NFC_MIN_WAVES
#define NFC_MIN_WAVES 2
#endif

/* Ring positions advance incrementally (add, compare) in the common kernel. While a stream is within 1024 samples
 * of its start or of the 32-bit wrap of its sample clock (once per 7 minutes at 10 MS/s) the positions are taken by
 * exact modulo instead, as the reference computes them: that variant is a kernel of its own, so that the common one
 * carries neither the modulo code nor branches between its history reads. Both kernels see the same launch; a
 * stream block is handled by exactly one of them, decided from the device state (the host only skips the launch of
 * the exact kernel when no stream of the launch can be near either point). */
__device__ __forceinline__ bool nfc_exact_span(uint32_t clock, uint32_t count)
{
   const uint32_t start = clock + 1u + 1024u; /* first sample clock of the launch, zone = [0, 2048) after the shift */
   const uint32_t untilWrap = 0u - start;
   return count != 0 && (start < 2048u || untilWrap < count);
}

/* input row of one stream slot for this launch; every field is wave-uniform (the slot is) */
struct NfcRow
{
   const uint8_t *data;
   uint32_t count;
};

typedef __attribute__((address_space(4))) const NfcWork NfcWorkConst;

__device__ __forceinline__ NfcRow nfc_row(const NfcLaunch &L, uint32_t slot)
{
   NfcRow row;
   row.data = nullptr;
   row.count = 0;

   if (slot >= L.firstSlot && slot < L.firstSlot + L.slotCount)
   {
      if (L.works)
      {
         /* constant address space: a uniform read through it is a scalar load */
         NfcWorkConst *w = (NfcWorkConst *)L.works + slot;
         row.data = w->data;
         row.count = w->count;
      }
      else
      {
         row.data = L.uniformBase + (uint64_t)(slot - L.firstSlot) * L.uniformPitch;
         row.count = L.uniformCount;
      }
   }

   return row;
}

/* Stage samples [base, base + TILE) of the 64 streams of a block into LDS as magnitudes, transposed: row r of the
 * tile is stream r, and one wave-wide load fetches 64 consecutive samples of one stream (512 B of IQ, 256 B of
 * magnitude). Rows are fetched NFC_STAGE_ROWS at a time, all loads of a batch in flight together; samples past the
 * end of a row read as zero (their lanes never consume them). S = floats per sample (2 IQ, 1 magnitude). */
#define NFC_STAGE_ROWS 16

template <uint32_t S>
__device__ __forceinline__ void nfc_stage_tile(const NfcLaunch &L, uint32_t block, uint32_t base, uint32_t lane, float *tile)
{
   const uint32_t idx = base + lane;

#pragma clang loop unroll(disable)
   for (uint32_t r0 = 0; r0 < NFC_LANES; r0 += NFC_STAGE_ROWS)
   {
      float re[NFC_STAGE_ROWS], im[NFC_STAGE_ROWS];
      uint32_t count[NFC_STAGE_ROWS];

#pragma unroll
      for (uint32_t j = 0; j < NFC_STAGE_ROWS; j++)
      {
         const NfcRow row = nfc_row(L, block * NFC_LANES + r0 + j);

         /* an empty row reads (and discards) from the ring storage, which is always there */
         const float *p = row.count ? (const float *)row.data : (const float *)L.rings;
         const uint32_t last = row.count ? row.count - 1u : 0u;
         const uint32_t at = idx < last ? idx : last;

         count[j] = row.count;

         if (S == 2)
         {
            const float2 iq = reinterpret_cast<const float2 *>(p)[at];
            re[j] = iq.x;
            im[j] = iq.y;
         }
         else
         {
            re[j] = p[at];
            im[j] = 0.0f;
         }
      }

#pragma unroll
      for (uint32_t j = 0; j < NFC_STAGE_ROWS; j++)
      {
         const float v = S == 2 ? nfc_iq_magnitude(re[j], im[j]) : re[j];
         if (TILE == NFC_LANES || lane < TILE)
            tile[(r0 + j) * TILE_PITCH + lane] = idx < count[j] ? v : 0.0f;
      }
   }
}

template <bool EXACT, bool FIXED>
__device__ __forceinline__ void nfc_demod_body(const NfcConfig *__restrict__ cfgPtr, const NfcLaunch &L, float *tile)
{
   const uint32_t lane = threadIdx.x;
   const uint32_t block = L.firstBlock + blockIdx.x;
   const uint32_t slot = block * NFC_LANES + lane;

   /* this lane's own row (per lane: the rows of a block may differ in length) */
   uint32_t mineCount = 0;

   if (slot >= L.firstSlot && slot < L.firstSlot + L.slotCount)
      mineCount = L.works ? L.works[slot].count : L.uniformCount;

   /* longest row of the block (wave-wide max) */
   uint32_t longest = mineCount;
   for (int off = 32; off > 0; off >>= 1)
   {
      uint32_t other = __shfl_xor(longest, off, 64);
      longest = other > longest ? other : longest;
   }

   longest = __builtin_amdgcn_readfirstlane(longest);

   if (longest == 0)
      return;

   NfcStreamState s = L.states[slot];

   /* the common kernel runs first and advances the clocks it decides on: a block it has taken must not be looked at
    * again by the exact kernel of the same launch (its clock may have moved into the zone meanwhile) */
   if (__any(mineCount != 0 && s.served == L.launchSeq) != 0)
      return;

   if ((L.forceExact != 0 || __any(nfc_exact_span(s.clock, mineCount)) != 0) != EXACT)
      return;

   NfcLaneMem mem;
   mem.ring = L.rings + (uint64_t)block * L.ringBlockFloats;
   mem.lane = lane;
   mem.exact = false;
   mem.linked = false;
   mem.flags = nullptr;
   mem.bytes = L.bytes + (uint64_t)slot * NFC_STREAM_BYTES;
   mem.sink = L.sink;
   mem.sinkCursor = L.sinkCtl;
   mem.sinkDropped = L.sinkCtl + 1;
   mem.sinkWords = L.sinkWords;
   mem.streamId = slot;
   mem.cold = L.cold + slot;
   mem.tables = cfgPtr;

   /* receive the state record here, once (see NFC_DRAIN) */
   NFC_DRAIN();

   for (uint32_t base = 0; base < longest; base += TILE)
   {
      if (L.uniformStride == 2)
         nfc_stage_tile<2>(L, block, base, lane, tile);
      else
         nfc_stage_tile<1>(L, block, base, lane, tile);

      __syncthreads();

      if (base < mineCount)
      {
         const uint32_t left = mineCount - base;
         const uint32_t n = left < TILE ? left : TILE;

         /* keep the ~100 configuration constants in the scalar cache instead of letting the compiler hoist
          * them out of the sample loop into (spilled) registers */
         const NfcConfigConst *cp = (const NfcConfigConst *)cfgPtr;

         if (FIXED)
         {
            /* periods, delays, ring offsets and filter weights are literals; thresholds and the enable mask are
             * the run-time part */
            NfcConfig cc;
            nfc_fixed_config(cc);
            cc.enabled = cp->enabled;
            cc.powerThreshold = cp->powerThreshold;
            cc.lowThreshold = cp->lowThreshold;
            cc.highThreshold = cp->highThreshold;
            for (int t = 0; t < 4; t++)
            {
               cc.corrThreshold[t] = cp->corrThreshold[t];
               cc.minDepth[t] = cp->minDepth[t];
               cc.maxDepth[t] = cp->maxDepth[t];
            }

            for (uint32_t k = 0; k < n; k++)
               nfc_step_as<EXACT>(cc, s, mem, tile[lane * TILE_PITCH + k]);
         }
         else
         {
            for (uint32_t k = 0; k < n; k++)
            {
#ifdef NFC_RELOAD_CONFIG
               asm volatile("" : "+s"(cp));
#endif
               nfc_step_as<EXACT>(*(const NfcConfig *)cp, s, mem, tile[lane * TILE_PITCH + k]);
            }
         }
      }

      __syncthreads();
   }

   if (mineCount)
   {
      s.served = L.launchSeq;
      L.states[slot] = s;
   }
}

#define NFC_DEMOD_KERNEL(name, exact, fixed, attrs)                                                        \
   __global__ __launch_bounds__(64) attrs void name(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) \
   {                                                                                                   \
      __shared__ float tile[NFC_LANES * TILE_PITCH];                                                   \
      nfc_demod_body<exact, fixed>(cfgPtr, L, tile);                                                   \
   }

/* exactly NFC_MIN_WAVES waves per SIMD: spilling state to scratch to reach a higher occupancy costs 5-10x */
#define NFC_PINNED __attribute__((amdgpu_waves_per_eu(NFC_MIN_WAVES, NFC_MIN_WAVES)))

/* any decodable sample rate */
NFC_DEMOD_KERNEL(nfc_demod_kernel, false, false, NFC_PINNED)
/* NFC_FIXED_SAMPLE_RATE with constants as literals */
NFC_DEMOD_KERNEL(nfc_demod_fixed_kernel, false, true, NFC_PINNED)
/* stream start / clock wrap variants (rare, not tuned: whatever occupancy the register allocator ends up with) */
NFC_DEMOD_KERNEL(nfc_demod_exact_kernel, true, false, )
NFC_DEMOD_KERNEL(nfc_demod_fixed_exact_kernel, true, true, )

/* out[i] = |iq[i]|, the formula the demodulation kernels apply while staging */
__global__ __launch_bounds__(256) void nfc_magnitude_kernel(const float2 *__restrict__ iq, float *__restrict__ out, uint64_t n)
{
   const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;

   for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
   {
      const float2 v = iq[i];
      out[i] = nfc_iq_magnitude(v.x, v.y);
   }
}

/* ------------------------------------------------------------------------------------------ */
/* adaptive resampler (SignalResamplingTask.cpp:168-226), one lane per buffer                   */
/* ------------------------------------------------------------------------------------------ */

#define NFC_RS_WINDOW 51      /* WINDOW */
#define NFC_RS_INTERVAL 255   /* RADIO_INTERVAL */
#define NFC_RS_TILE 32        /* samples staged per pass */
#define NFC_RS_RING 96        /* per-lane sample window in LDS: >= WINDOW + TILE, multiple of TILE */
#define NFC_RS_PITCH 97

/* The running mean is a sequential fp32 sum per buffer (add the sample entering the centred window, subtract the one
 * leaving it, in that order), so a buffer is one lane's work; 64 buffers share a wave. Input rows are staged 32
 * samples at a time (two rows per load, 128 B each) into a per-lane ring in LDS that always holds the 51-sample
 * window of the sample being decided; a sample is decided once the 25 samples after it are there. */
__global__ __launch_bounds__(64) void nfc_resample_radio_kernel(const float *__restrict__ in, uint64_t pitchFloats, uint32_t nBuffers, uint32_t n,
                                                               float *__restrict__ out, uint64_t outPitchFloats, uint32_t capacityPairs,
                                                               uint32_t *__restrict__ counts)
{
   __shared__ float ring[NFC_LANES * NFC_RS_PITCH];

   const uint32_t lane = threadIdx.x;
   const uint32_t buffer = blockIdx.x * NFC_LANES + lane;
   const bool mine = buffer < nBuffers;

   float *dst = out + (uint64_t)buffer * outPitchFloats;
   uint32_t count = 0;

   auto put = [&](float value, float offset) {
      if (mine && count < capacityPairs)
         reinterpret_cast<float2 *>(dst)[count] = make_float2(value, offset); /* rows are 8-byte aligned (pitch % 8 == 0) */
      count++;
   };

   const float *window = ring + lane * NFC_RS_PITCH;

   float avrg = 0.0f, last = 0.0f;
   const float filter = 0.005f; /* THRESHOLD */

   int32_t i = 0, c = 0, p = -1;
   uint32_t posI = 0, posA = NFC_RS_WINDOW / 2, posR = NFC_RS_RING - (NFC_RS_WINDOW / 2) - 1; /* ring columns of i, a, r */

   const uint32_t col = lane % NFC_RS_TILE;
   const uint32_t half = lane / NFC_RS_TILE;

   for (uint32_t base = 0; base < n; base += NFC_RS_TILE)
   {
      const uint32_t idx = base + col;
      const uint32_t at = (base % NFC_RS_RING) + col;

#pragma clang loop unroll(disable)
      for (uint32_t r0 = 0; r0 < NFC_LANES; r0 += 32)
      {
         float v[16];

#pragma unroll
         for (uint32_t j = 0; j < 16; j++)
         {
            const uint32_t row = r0 + 2 * j + half;
            const uint32_t rb = blockIdx.x * NFC_LANES + row;
            const bool ok = rb < nBuffers && idx < n;
            v[j] = ok ? in[(uint64_t)rb * pitchFloats + idx] : 0.0f;
         }

#pragma unroll
         for (uint32_t j = 0; j < 16; j++)
            ring[(r0 + 2 * j + half) * NFC_RS_PITCH + at] = v[j];
      }

      __syncthreads();

      const uint32_t filled = base + NFC_RS_TILE < n ? base + NFC_RS_TILE : n;

      if (base == 0)
      {
         /* "initialize average" and "always store first sample" */
         for (uint32_t k = 0; k < NFC_RS_WINDOW / 2; k++)
            avrg += window[k];

         last = window[0];
         put(window[0], 0.0f);
      }

      /* decide every sample whose window is complete (all of them once the buffer has been read to its end) */
      const int32_t stop = filled == n ? (int32_t)n : (int32_t)filled - NFC_RS_WINDOW / 2;

      for (; i < stop; ++i, ++p)
      {
         const float value = window[posI];

         if ((uint32_t)(i + NFC_RS_WINDOW / 2) < n)
            avrg += window[posA];

         if (i - NFC_RS_WINDOW / 2 - 1 >= 0)
            avrg -= window[posR];

         const float stdev = fabsf(value - (avrg / (float)NFC_RS_WINDOW));

         if (stdev > filter || (i - c) >= NFC_RS_INTERVAL)
         {
            if (stdev > filter && c < p)
               put(last, (float)p);

            put(value, (float)i);

            c = i;
         }

         last = value;

         posI = posI + 1 == NFC_RS_RING ? 0 : posI + 1;
         posA = posA + 1 == NFC_RS_RING ? 0 : posA + 1;
         posR = posR + 1 == NFC_RS_RING ? 0 : posR + 1;
      }

      __syncthreads();
   }

   if (c < p)
      put(last, (float)p);

   if (mine)
      counts[buffer] = count;
}

__global__ __launch_bounds__(64) void nfc_init_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, uint32_t keepFrontEnd)
{
   const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;

   if (idx >= L.slotCount)
      return;

   const uint32_t slot = L.firstSlot + idx;
   const uint32_t block = slot / NFC_LANES;
   const uint32_t lane = slot % NFC_LANES;

   const NfcConfig &cfg = *cfgPtr;

   NfcStreamState s = L.states[slot];
   NfcStreamCold cold;
   nfc_state_init(cfg, s, cold, keepFrontEnd != 0);
   L.states[slot] = s;
   L.cold[slot] = cold;

   float *ring = L.rings + (uint64_t)block * L.ringBlockFloats + lane;

   /* the reference's initialize() leaves the sample history alone and clears every modulation buffer */
   const uint32_t from = keepFrontEnd ? 4 * NFC_HIST : 0;
   const uint32_t total = L.ringBlockFloats / NFC_LANES;

   for (uint32_t i = from; i < total; i++)
      ring[i * NFC_LANES] = 0.0f;
}

/* ------------------------------------------------------------------------------------------ */
/* time-parallel path (nfc_scan.h): scan -> seams -> windows -> lanes -> windowed decode -> chain -> finish */
/* ------------------------------------------------------------------------------------------ */

/* run-time part of the configuration on top of the compiled-in table (the path is only taken at that sample rate) */
__device__ __forceinline__ void nfc_fixed_runtime_config(const NfcConfig *cfgPtr, NfcConfig &cc)
{
   const NfcConfigConst *cp = (const NfcConfigConst *)cfgPtr;

   nfc_fixed_config(cc);
   cc.enabled = cp->enabled;
   cc.powerThreshold = cp->powerThreshold;
   cc.lowThreshold = cp->lowThreshold;
   cc.highThreshold = cp->highThreshold;
   for (int t = 0; t < 4; t++)
   {
      cc.corrThreshold[t] = cp->corrThreshold[t];
      cc.minDepth[t] = cp->minDepth[t];
      cc.maxDepth[t] = cp->maxDepth[t];
   }
}

/* The scan kernel: one lane per chunk of a stream, 64 chunks per wave. Per step the wave fetches 64 consecutive samples
 * of each of its 64 chunks with one coalesced row load each (512 B of IQ / 256 B of magnitude; row descriptors are
 * parked in LDS once and read back through the scalar unit), converts to magnitude and parks the tile transposed in LDS;
 * every lane then walks its own row through the exact front end and the tile records. Nothing is written per sample:
 * a 24-byte record per 64 samples, a 32-byte front-end state per 512.
 * Bound: HBM read (8 B per IQ sample, plus the warm-up overlap warmSamples / chunkSamples). */
#define NFC_SCAN_ROWS 16
#define NFC_SCAN_PITCH (NFC_SCAN_TILE + 1)

/* S = floats per sample (2 IQ, 1 magnitude) */
template <uint32_t S>
__device__ __forceinline__ void nfc_scan_body(const NfcConfig *__restrict__ cfgPtr, const NfcScanArgs &A, float *tile, uint32_t *rows)
{
   const uint32_t lane = threadIdx.x;
   const uint32_t listed = blockIdx.x * NFC_LANES + lane;
   const bool mine = listed < A.nChunks + A.nChunksMore;

   NfcScanChunk ch;
   ch.job = 0;
   ch.index = 0;
   if (mine)
      ch = listed < A.nChunks ? A.chunks[listed] : A.chunksMore[listed - A.nChunks];

   /* a chunk on the repair list is walked from the true end state of the chunk before, without warm-up */
   const bool repair = (ch.index & NFC_CHUNK_REPAIR) != 0;
   const bool envelopeOnly = repair && (ch.index & NFC_CHUNK_ENVELOPE) != 0; /* only the envelope tracker is walked again */
   ch.index &= ~(NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE);

   const NfcScanJob *job = A.jobs + ch.job;
   const uint32_t g = job->firstChunk + ch.index; /* seam / chunk record */

   const uint32_t count = mine ? job->count : 0u;
   const uint32_t L = A.params.chunkSamples, WU = A.params.warmSamples;

   const uint32_t start = ch.index * L;
   const uint32_t end = mine ? (start + L < count ? start + L : count) : 0u;
   const uint32_t walkFrom = (ch.index == 0 || repair) ? start : start - WU;
   const int32_t origin = (int32_t)start - (int32_t)WU; /* stream position of this lane's row at step 0 (may be negative) */
   const uint32_t reseedAt = walkFrom + WU / 3 / NFC_SCAN_TILE * NFC_SCAN_TILE;

   /* row descriptor (this lane's; the other lanes read it with v_readlane): where the row is at step 0, and the steps'
    * sample range the walk covers: [fromRel, endRel) */
   const uint64_t rowBase = (uint64_t)(mine ? job->data : (const uint8_t *)A.tileStats) + (int64_t)origin * (int64_t)(S * 4u);
   const uint32_t rowLo = (uint32_t)rowBase, rowHi = (uint32_t)(rowBase >> 32);
   const uint32_t rowFrom = walkFrom - (uint32_t)origin; /* origin <= walkFrom */
   uint32_t rowEnd = mine ? end - (uint32_t)origin : 0u;  /* (a second walk that meets the first one's trajectory ends early) */
   (void)rows;

   NfcConfig cc;
   nfc_fixed_runtime_config(cfgPtr, cc);

   const NfcStreamState *from = nullptr;
   uint32_t clockBase = 0;
   if (mine)
   {
      const NfcStreamState *st = A.states + job->slot;
      clockBase = st->clock;
      if (ch.index == 0)
         from = st;
   }

   NfcScanLane w;
   __builtin_memset(&w, 0, sizeof(w));
   bool begun = false;

   NfcScanSeam seam;
   __builtin_memset(&seam, 0, sizeof(seam));

   const uint32_t myFrom = rowFrom;
   uint32_t myEnd = rowEnd;
   bool merged = false;

   /* The rows of the next step are fetched into registers while this step's tile is walked: one memory latency per step,
    * hidden behind the walk (64 row loads in flight per wave). */
   float re[NFC_LANES], im[NFC_LANES];

   auto fetch = [&](uint32_t rel) {
#pragma unroll
      for (uint32_t q = 0; q < NFC_LANES; q++)
      {
         /* lane q's descriptor, into scalar registers */
         const uint32_t lo = __builtin_amdgcn_readlane(rowLo, q);
         const uint32_t hi = __builtin_amdgcn_readlane(rowHi, q);
         const uint32_t fromRel = __builtin_amdgcn_readlane(rowFrom, q);
         const uint32_t endRel = __builtin_amdgcn_readlane(rowEnd, q);

         /* Does the row have anything at this step (uniform)? The load is issued either way - from the first word of the
          * tile records where it has not - so that the 64 of a step are in flight together: behind a branch each would
          * wait for the one before (seen on the second walks, which run a wave or two per CU: a memory latency per row). */
         const bool has = rel + NFC_SCAN_TILE > fromRel && rel < endRel;

         typedef __attribute__((address_space(1))) const float GlobalFloat;
         const uint64_t row = ((uint64_t)hi << 32) | lo;

         uint32_t at = rel + lane;
         at = at < fromRel ? fromRel : at;
         at = at >= endRel ? endRel - 1u : at;

         if (S == 2)
         {
            GlobalFloat *p = has ? (GlobalFloat *)row + 2u * at : (GlobalFloat *)A.tileStats;
            const float vx = p[0], vy = p[1];
            re[q] = has ? vx : 0.0f;
            im[q] = has ? vy : 0.0f;
         }
         else
         {
            GlobalFloat *p = has ? (GlobalFloat *)row + at : (GlobalFloat *)A.tileStats;
            const float v = *p;
            re[q] = has ? v : 0.0f;
            im[q] = 0.0f;
         }
      }
   };

   fetch(0);

   for (uint32_t rel = 0; rel < WU + L; rel += NFC_SCAN_TILE)
   {
      /* (second walks end where they meet the first one's trajectory) */
      if (__ballot(rel < myEnd) == 0ull)
         break;

      /* ---- park the fetched rows in LDS as magnitudes, transposed: tile row q = lane q's 64 samples of this step ---- */
#pragma unroll
      for (uint32_t q = 0; q < NFC_LANES; q++)
         tile[q * NFC_SCAN_PITCH + lane] = S == 2 ? nfc_iq_magnitude(re[q], im[q]) : re[q];

      __syncthreads();

      if (rel + NFC_SCAN_TILE < WU + L)
         fetch(rel + NFC_SCAN_TILE);

      /* ---- walk: this lane's 64 samples ---- */
      if (rel + NFC_SCAN_TILE > myFrom && rel < myEnd)
      {
         const uint32_t pos = (uint32_t)(origin + (int32_t)rel); /* stream position of the tile's first sample (>= 0 here) */
         const uint32_t n = myEnd - rel < NFC_SCAN_TILE ? myEnd - rel : NFC_SCAN_TILE;
         const float *row = tile + lane * NFC_SCAN_PITCH;

         if (envelopeOnly)
         {
            /* The second walk of a chunk whose envelope tracker alone had started wrong (nfc_seams_check): envelope and
             * pulse counter from the true start; what depends on them is put right - the stored points' two fields, the
             * tiles' envelope extremes - until the walk meets the first one's trajectory. */
            if (!begun)
            {
               seam = A.seams[g];
               w.fe.clock = clockBase + pos;
               w.fe.env = seam.start.env;
               w.fe.pulseFilter = seam.start.pulseFilter;
               begun = true;

               /* the point stored where the chunk begins carries the true start too (the full second walk rewrites it;
                * this one left the first walk's guess there) */
               if (pos == start && (start % NFC_SCAN_POINT) == 0)
               {
                  NfcScanPoint &first = A.points[job->firstPoint + start / NFC_SCAN_POINT];
                  first.env = seam.start.env;
                  first.pulseFilter = seam.start.pulseFilter;
               }
            }

            if (pos > start && (pos % NFC_SCAN_POINT) == 0)
            {
               NfcScanPoint &stored = A.points[job->firstPoint + pos / NFC_SCAN_POINT];

               if (nfc_bits(stored.env) == nfc_bits(w.fe.env) && stored.pulseFilter == w.fe.pulseFilter)
               {
                  seam.end = A.seams[g].end; /* the first walk's end stands */
                  merged = true;
                  myEnd = rel;
                  rowEnd = rel;
               }
               else
               {
                  stored.env = w.fe.env;
                  stored.pulseFilter = w.fe.pulseFilter;
               }
            }

            if (!merged)
            {
               float lo = NFC_SCAN_BIG, hi = -NFC_SCAN_BIG;

               for (uint32_t k = 0; k < n; k++)
               {
                  ++w.fe.clock;
                  ++w.fe.pulseFilter;
                  nfc_envelope_step(cc, w.fe.clock, w.fe.pulseFilter, w.fe.env, row[k]);
                  lo = w.fe.env < lo ? w.fe.env : lo;
                  hi = w.fe.env > hi ? w.fe.env : hi;
               }

               NfcScanTile &stat = A.tileStats[job->firstTile + pos / NFC_SCAN_TILE];
               stat.envmin = lo;
               stat.envmax = hi;
               stat.bits |= NFC_TILE_REWALKED;

               seam.end.env = w.fe.env;
               seam.end.pulseFilter = w.fe.pulseFilter;
            }
         }
         else
         {
         if (!begun)
         {
            if (repair)
               nfc_scan_resume(w, A.seams[g].start, A.seams[g].start.edgeTime, clockBase + pos);
            else
            {
               /* guess for the envelope and the average: mean of the first tile of the walk */
               float first = 0.0f;
               for (uint32_t k = 0; k < n; k++)
                  first += row[k];
               first = first / (float)n;

               nfc_scan_begin(w, from, clockBase + pos, first);
            }
            begun = true;
         }

         if (ch.index != 0 && !repair && pos == reseedAt)
            nfc_scan_reseed(w);

         if (pos == start)
            nfc_scan_point(w, seam.start);

         if (pos >= start && (pos % NFC_SCAN_POINT) == 0)
         {
            NfcScanPoint &stored = A.points[job->firstPoint + pos / NFC_SCAN_POINT];

            /* a second walk: has it met the first one's trajectory? Then the rest of the chunk stands as recorded
             * (nfc_scan_merged), up to the edge time the first walk may not have trusted */
            if (repair && pos > start)
            {
               NfcScanPoint here;
               nfc_scan_point(w, here);

               if (nfc_scan_merged(here, stored))
               {
                  const uint32_t atMerge = stored.edgeTime;

                  for (uint32_t q = pos + NFC_SCAN_POINT; q < end; q += NFC_SCAN_POINT)
                     nfc_scan_adopt(A.points[job->firstPoint + q / NFC_SCAN_POINT], atMerge, w.fe.edgeTime);

                  seam.end = A.seams[g].end;
                  nfc_scan_adopt(seam.end, atMerge, w.fe.edgeTime);

                  stored = here;
                  merged = true;
                  myEnd = rel;
                  rowEnd = rel; /* nothing more to fetch for this row */
               }
            }

            if (!merged)
               nfc_scan_point(w, stored);
         }

         if (!merged)
         {

         /* (whole tiles - all but the last of a stream - with a fixed trip count) */
         if (n == NFC_SCAN_TILE)
         {
#pragma unroll 8
            for (uint32_t k = 0; k < NFC_SCAN_TILE; k++)
               (void)nfc_scan_sample(cc, w, row[k]);
         }
         else
         {
            for (uint32_t k = 0; k < n; k++)
               (void)nfc_scan_sample(cc, w, row[k]);
         }

         NfcScanTile stat;
         nfc_scan_tile_end(w, stat);

         if (repair)
            stat.bits |= NFC_TILE_REWALKED;

         if (pos >= start)
            A.tileStats[job->firstTile + pos / NFC_SCAN_TILE] = stat;
         }
         }
      }

      __syncthreads();
   }

   if (mine)
   {
      if (envelopeOnly)
      {
         if (begun)
            A.seams[g] = seam; /* (the first walk's record with the tracker's end put right, or untouched after a merge) */
      }
      else
      {
         if (begun && !merged)
            nfc_scan_point(w, seam.end);
         if (repair)
            seam.start = A.seams[g].start; /* as the seam check set it */
         A.seams[g] = seam;
      }
   }
}

__global__ __launch_bounds__(64) void nfc_scan_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A)
{
   __shared__ float tile[NFC_LANES * NFC_SCAN_PITCH];
   __shared__ uint32_t rows[NFC_LANES * 4];

   if (A.stride == 2)
      nfc_scan_body<2>(cfgPtr, A, tile, rows);
   else
      nfc_scan_body<1>(cfgPtr, A, tile, rows);
}

/* The front-end planes (NfcScanArgs::planes: {filtered, envelope, deviation, average} after every sample, 16 B) for the wave
 * decoder: a walk of the front end alone over every chunk from its verified start state, a lane per chunk, the rows fetched
 * and transposed like the scan's.
 * Round 5. Until now this was the scan's body with a store per sample and lane: sixty-four 16-byte pieces per store
 * instruction, each in a row of its own half a megabyte from the next, which the cache had to piece together into lines
 * (35 ms for config 5's 69 GB, 2 TB/s; the same stores sent past the cache - nontemporal - took 213 ms: the merging is what
 * made it bearable). Here four samples of every lane go through LDS and leave as 64-byte pieces - a store instruction writes
 * sixteen rows' four records each -, and the walk is the decoder's front end and nothing else (the tile extremes, grid test
 * and edge bookkeeping of nfc_scan_sample are the scan's business). */
#ifndef NFC_PLANES_GROUP
#define NFC_PLANES_GROUP 4u
#endif

template <uint32_t S>
__device__ __forceinline__ void nfc_planes_body(const NfcConfig *__restrict__ cfgPtr, const NfcScanArgs &A, float *tile, float4 *stage, uint64_t *rowOut, uint32_t *rowN)
{
   const uint32_t lane = threadIdx.x;
   const uint32_t listed = blockIdx.x * NFC_LANES + lane;
   const uint32_t per = A.planesPerChunk ? A.planesPerChunk : 1u;
   const bool mine = listed / per < A.nChunks;

   NfcScanChunk ch;
   ch.job = 0;
   ch.index = 0;
   if (mine)
      ch = A.chunks[listed / per];
   ch.index &= ~(NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE);

   /* (an entry that is a chunk walked by several lanes, a piece each: NfcScanArgs::planesPerChunk) */
   if (A.planesPerChunk)
      ch.index = ch.index * per + listed % per;

   const NfcScanJob *job = A.jobs + ch.job;
   const uint32_t count = mine ? job->count : 0u;
   const uint32_t L = A.planesPiece ? A.planesPiece : A.params.chunkSamples;
   const uint32_t start = ch.index * L;
   const uint32_t end = mine ? (start + L < count ? start + L : count) : 0u;
   const uint32_t mySamples = end > start ? end - start : 0u;

   /* row descriptor (this lane's; the other lanes read it with v_readlane): the chunk's first sample and its length */
   const uint64_t rowBase = (uint64_t)(mySamples ? job->data : (const uint8_t *)A.tileStats) + (uint64_t)start * (uint64_t)(S * 4u);
   const uint32_t rowLo = (uint32_t)rowBase, rowHi = (uint32_t)(rowBase >> 32);

   NfcConfig cc;
   nfc_fixed_runtime_config(cfgPtr, cc);

   NfcScanLane w;
   __builtin_memset(&w, 0, sizeof(w));

   /* from the chunk's verified start - or, a small submission (NfcScanArgs::planesPiece): from the point stored every 512
    * samples, which the second walks have left true like the seams: more lanes, each a shorter walk */
   if (mySamples)
   {
      const NfcScanPoint &from = A.planesPiece ? A.points[job->firstPoint + start / NFC_SCAN_POINT] : A.seams[job->firstChunk + ch.index].start;
      nfc_scan_resume(w, from, from.edgeTime, A.states[job->slot].clock + start);
   }

   float4 *const outBase = reinterpret_cast<float4 *>(A.planes) + ((uint64_t)job->firstTile * NFC_SCAN_TILE + start);

   float re[NFC_LANES], im[NFC_LANES];

   auto fetch = [&](uint32_t rel) {
#pragma unroll
      for (uint32_t q = 0; q < NFC_LANES; q++)
      {
         const uint32_t lo = __builtin_amdgcn_readlane(rowLo, q);
         const uint32_t hi = __builtin_amdgcn_readlane(rowHi, q);
         const uint32_t endRel = __builtin_amdgcn_readlane(mySamples, q);
         const bool has = rel < endRel; /* (uniform; the load is issued either way: the 64 of a step in flight together) */

         typedef __attribute__((address_space(1))) const float GlobalFloat;
         const uint64_t row = ((uint64_t)hi << 32) | lo;

         uint32_t at = rel + lane;
         at = (has && at >= endRel) ? endRel - 1u : at;

         if (S == 2)
         {
            GlobalFloat *p = has ? (GlobalFloat *)row + 2u * at : (GlobalFloat *)A.tileStats;
            const float vx = p[0], vy = p[1];
            re[q] = has ? vx : 0.0f;
            im[q] = has ? vy : 0.0f;
         }
         else
         {
            GlobalFloat *p = has ? (GlobalFloat *)row + at : (GlobalFloat *)A.tileStats;
            const float v = *p;
            re[q] = has ? v : 0.0f;
            im[q] = 0.0f;
         }
      }
   };

   fetch(0);

   for (uint32_t rel = 0; rel < L; rel += NFC_SCAN_TILE)
   {
      if (__ballot(rel < mySamples) == 0ull)
         break;

#pragma unroll
      for (uint32_t q = 0; q < NFC_LANES; q++)
         tile[q * NFC_SCAN_PITCH + lane] = S == 2 ? nfc_iq_magnitude(re[q], im[q]) : re[q];

      /* where this lane's records of the step go, and how many there are (for the lanes that will store them) */
      const uint32_t n = rel < mySamples ? (mySamples - rel < NFC_SCAN_TILE ? mySamples - rel : NFC_SCAN_TILE) : 0u;
      rowOut[lane] = n ? (uint64_t)(outBase + rel) : 0ull;
      rowN[lane] = n;

      __syncthreads();

      if (rel + NFC_SCAN_TILE < L)
         fetch(rel + NFC_SCAN_TILE);

      const float *row = tile + lane * NFC_SCAN_PITCH;

      for (uint32_t k0 = 0; k0 < NFC_SCAN_TILE; k0 += NFC_PLANES_GROUP)
      {
#pragma unroll
         for (uint32_t j = 0; j < NFC_PLANES_GROUP; j++)
         {
            if (k0 + j < n)
            {
               ++w.fe.clock;
               ++w.fe.pulseFilter;
               const NfcNow now = nfc_front_end_core(cc, w.fe, row[k0 + j]);
               stage[lane * NFC_PLANES_GROUP + j] = make_float4(now.filt, w.fe.env, w.fe.mdev, w.fe.avg);
            }
         }

         __syncthreads();

         /* a store instruction: sixteen rows, the four records of each (64 B contiguous per row) */
#pragma unroll
         for (uint32_t i = 0; i < NFC_LANES * NFC_PLANES_GROUP / NFC_LANES; i++)
         {
            const uint32_t r = i * (NFC_LANES / NFC_PLANES_GROUP) + lane / NFC_PLANES_GROUP;
            const uint32_t piece = lane % NFC_PLANES_GROUP;
            float4 *to = (float4 *)rowOut[r];

            if (to && k0 + piece < rowN[r])
               to[k0 + piece] = stage[r * NFC_PLANES_GROUP + piece];
         }

         __syncthreads();
      }
   }
}

__global__ __launch_bounds__(64) void nfc_scan_planes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A)
{
   __shared__ float tile[NFC_LANES * NFC_SCAN_PITCH];
   __shared__ float4 stage[NFC_LANES * NFC_PLANES_GROUP];
   __shared__ uint64_t rowOut[NFC_LANES];
   __shared__ uint32_t rowN[NFC_LANES];

   if (A.stride == 2)
      nfc_planes_body<2>(cfgPtr, A, tile, stage, rowOut, rowN);
   else
      nfc_planes_body<1>(cfgPtr, A, tile, stage, rowOut, rowN);
}

/* The chunks whose start state changed after the walk that writes the planes was started (NfcScanArgs::planesStale): listed for
 * a walk of their own. `all`: the submission's chunk table, chunk after chunk as the seam records lie. */
__global__ __launch_bounds__(256) void nfc_planes_stale_kernel(NfcScanArgs A, const NfcScanChunk *all, uint32_t nAll, NfcScanChunk *out, uint32_t *count)
{
   const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;

   if (g >= nAll || !A.planesStale[g])
      return;

   out[atomicAdd(count, 1u)] = all[g];
}

/* one thread per job: seams, then windows (the two are cheap and sequential per stream) */
__global__ __launch_bounds__(64) void nfc_seams_kernel(NfcScanArgs A, uint32_t first)
{
   const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;

   if (j >= A.nJobs)
      return;

   NfcScanJob job = A.jobs[j];

   if (first)
      job.passes = 0;

   if (!(job.status & NFC_JOB_INVALID))
      nfc_seams_check(job, j, A.seams, A.chunkEdge, A.states[job.slot].edgeTime, A.repairs, A.repairCount, A.points, A.params.chunkSamples, A.repairsEnv, A.repairEnvCount,
                      A.planesStale);

   A.jobs[j] = job;
}

/* one thread per tile: the tile tests. Grid: x over the jobs, y over the tiles of a job (256 to a block) (round 5; a thread used
 * to find its job by a binary search of the job table - twelve dependent trips to memory for config 5 - : 3.8 ms for its 67 M
 * tiles); nTilesMost: tiles of the longest job */
__global__ __launch_bounds__(256) void nfc_tiles_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, uint32_t nTilesMost)
{
   const uint32_t lo = blockIdx.x;

   if (lo >= A.nJobs)
      return;

   NfcConfig cc;
   nfc_fixed_runtime_config(cfgPtr, cc);

   const uint32_t first = A.jobs[lo].firstTile;
   const uint32_t nTiles = (A.jobs[lo].count + NFC_SCAN_TILE - 1u) / NFC_SCAN_TILE;

   /* (the host keeps grid.y inside the 65535 blocks a grid may have in y: a job of more than 2^24 tiles strides) */
   for (uint32_t inJob = blockIdx.y * blockDim.x + threadIdx.x; inJob < nTiles && inJob < nTilesMost; inJob += gridDim.y * blockDim.x)
   {
      const uint32_t i = first + inJob;
      const uint32_t flags = nfc_tile_flags(cc, A.params, A.tileStats + first, inJob);

      A.tiles[i] = flags;

      /* Off the capture grid a running sum depends on everything that was ever added to it: no lane that starts inside the
       * stream can be in the decoder's state. The carry lane is (it continues the stream's own sums), so it decodes such a
       * stream alone; the lane-per-window kernels have no way of walking the sums and leave it to the sequential ones. */
      if (flags & NFC_TILE_OFFGRID)
         atomicOr(&A.jobs[lo].status, A.params.offGridAlone ? (NFC_JOB_ALONE | NFC_JOB_OFFGRID_SEEN) : NFC_JOB_OFFGRID);

      if ((flags & NFC_TILE_BUSY) && !(flags & NFC_TILE_DARK))
         atomicAdd(&A.jobs[lo].busyTiles, 1u);
   }
}

/* One wave per job: retire / dark marks and the windows, 64 tiles per step (nfc_scan.hpp: nfc_group_*). Every lane of
 * the wave holds the same masks and the same placer state; the lanes share the stores of a window record (NFC_WINDOW_STORE). */
__device__ __forceinline__ uint32_t nfc_windows_place(const NfcScanJob &job, uint32_t j, const uint32_t *__restrict__ t, uint32_t nTiles, NfcWindow *out,
                                                      uint32_t room, bool write)
{
   const uint32_t lane = threadIdx.x;
   const uint32_t groups = (nTiles + 63u) / 64u;

   NfcWindowPlacer placer {0u, 0u};
   uint64_t busyBefore = ~0ull;

   for (uint32_t g0 = 0; g0 < groups; g0 += 4u)
   {
      uint32_t f[4];

      for (uint32_t k = 0; k < 4u; k++)
      {
         const uint32_t i = (g0 + k) * 64u + lane;
         f[k] = i < nTiles ? t[i] : 0u;
      }

      for (uint32_t k = 0; k < 4u && g0 + k < groups; k++)
      {
         const uint32_t g = g0 + k;
         const uint32_t i = g * 64u + lane;
         const bool exists = i < nTiles;

         const uint64_t busy = __ballot(exists && (f[k] & NFC_TILE_BUSY) != 0u);
         const uint64_t cluster = __ballot(exists && nfc_tile_may_start(i, job.count) && nfc_group_cluster(busyBefore, busy, lane));
         const uint64_t cut = __ballot(exists && !(f[k] & (NFC_TILE_RETIRE_OK | NFC_TILE_DARK)) && nfc_tile_may_cut(i, job.count));

         if (cluster | cut)
            nfc_group_place(placer, job, j, out, room, g, cluster, cut, write);

         busyBefore = busy;
      }
   }

   nfc_windows_close(placer, job, j, out, room, write);
   return placer.n;
}

__global__ __launch_bounds__(64) void nfc_windows_kernel(NfcScanArgs A)
{
   const uint32_t j = blockIdx.x;
   const uint32_t lane = threadIdx.x;

   if (j >= A.nJobs)
      return;

   NfcScanJob job = A.jobs[j];

   job.status &= ~NFC_JOB_OVERFLOW;

   if (job.status & NFC_JOB_INVALID)
   {
      job.windows = 0;
      if (lane == 0u)
         A.jobs[j] = job;
      return;
   }

   const uint32_t nTiles = (job.count + NFC_SCAN_TILE - 1) / NFC_SCAN_TILE;
   const uint32_t groups = (nTiles + 63u) / 64u;
   uint32_t *t = A.tiles + job.firstTile;

   /* backwards: where a lane may retire, and how much dark signal lies ahead */
   {
      uint64_t blockedNext = ~0ull;
      uint32_t carry = 0u;

      for (uint32_t done = 0; done < groups; done += 4u)
      {
         uint32_t f[4];

         for (uint32_t k = 0; k < 4u; k++)
         {
            const uint32_t g = groups - 1u - done - k; /* wraps for k beyond the first group: i >= nTiles then */
            const uint32_t i = g * 64u + lane;
            f[k] = (done + k < groups && i < nTiles) ? t[i] : 0u;
         }

         for (uint32_t k = 0; k < 4u && done + k < groups; k++)
         {
            const uint32_t g = groups - 1u - done - k;
            const uint32_t i = g * 64u + lane;
            const bool exists = i < nTiles;

            const uint64_t blocked = __ballot(!exists || (f[k] & NFC_TILE_BUSY) != 0u);
            const uint64_t dark = __ballot(exists && (f[k] & NFC_TILE_DARK) != 0u);
            const bool full = g * 64u + 64u <= nTiles;

            const uint32_t run = nfc_group_dark_run(dark, full, lane, carry);
            const bool ok = nfc_group_retire_ok(blocked, blockedNext, lane);

            if (exists)
               t[i] = (f[k] & 0xFFFFu & ~NFC_TILE_RETIRE_OK) | (ok ? NFC_TILE_RETIRE_OK : 0u) | (run << NFC_TILE_DARK_RUN_SHIFT);

            carry = nfc_group_dark_run(dark, full, 0u, carry);
            blockedNext = blocked;
         }
      }
   }

   __threadfence_block();
   __syncthreads();

   /* count, reserve, fill: the speculative windows of a job are contiguous and ordered */
   /* a short stream is decoded by its carry lane alone (NfcScanParams::soloSamples): one pass, nothing to speculate on */
   const bool solo = job.count <= A.params.soloSamples || (job.status & NFC_JOB_ALONE) != 0u;
   const uint32_t need = solo ? 0u : nfc_windows_place(job, j, t, nTiles, nullptr, 0u, false);

   uint32_t first = 0u;
   if (lane == 0u)
      first = atomicAdd(A.windowCount, need);
   first = (uint32_t)__shfl((int)first, 0, 64);

   job.firstWindow = A.firstWindowSlot + first;
   job.windows = need;

   if (first + need > A.windowRoom)
   {
      job.status |= NFC_JOB_OVERFLOW;
      job.windows = 0;
   }
   else if (need)
      (void)nfc_windows_place(job, j, t, nTiles, A.windows + job.firstWindow, need, true);

   if (lane == 0u)
      A.jobs[j] = job;
}

/* Prepare lanes. Carry lanes (window index == job index, one wave per job): a copy of the stream's slot, so that the
 * stream itself stays untouched until the submission is settled. Speculative lanes: scanned front end + assumed carry. */
__global__ __launch_bounds__(64) void nfc_carry_lanes_kernel(NfcScanArgs A, NfcLaunch real, NfcLaunch lanes, uint32_t pass)
{
   const uint32_t j = blockIdx.x;
   const uint32_t t = threadIdx.x;

   if (j >= A.nJobs)
      return;

   const NfcScanJob *job = A.jobs + j;

   /* later passes: only the carry lanes the chain kernel sent on past a lane they had handed over to */
   const uint32_t noHand = pass ? A.windows[j].noHand : 0u;

   if (pass && !A.windows[j].rerun)
   {
      if (t == 0)
         A.works[j].count = 0;
      return;
   }
   const uint32_t from = job->slot, to = j;

   const float *src = real.rings + (uint64_t)(from / NFC_LANES) * real.ringBlockFloats + (from % NFC_LANES);
   float *dst = lanes.rings + (uint64_t)(to / NFC_LANES) * lanes.ringBlockFloats + (to % NFC_LANES);

   for (uint32_t i = t; i < real.ringBlockFloats / NFC_LANES; i += NFC_LANES)
      dst[(uint64_t)i * NFC_LANES] = src[(uint64_t)i * NFC_LANES];

   for (uint32_t i = t; i < NFC_STREAM_BYTES / 4; i += NFC_LANES)
      ((uint32_t *)(lanes.bytes + (uint64_t)to * NFC_STREAM_BYTES))[i] = ((const uint32_t *)(real.bytes + (uint64_t)from * NFC_STREAM_BYTES))[i];

   if (t == 0)
   {
      NfcStreamState s = real.states[from];
      NfcStreamCold cold = real.cold[from];

      cold.frameHead = 0;
      cold.frameTail = 0;
      cold.emitOwn = 0; /* (a lane's notes: NfcStreamCold) */
      cold.waitFlags = 0;
      cold.waitUsed[0] = cold.waitUsed[1] = cold.waitUsed[2] = cold.waitUsed[3] = 0;

      lanes.states[to] = s;
      lanes.cold[to] = cold;

      NfcWindow w;
      __builtin_memset(&w, 0, sizeof(w));
      w.job = j;
      w.verify = 0xFFFFFFFFu;
      w.noHand = noHand;
      nfc_carry_take(w.carry, s, cold);
      w.want = w.carry;
      A.windows[to] = w;

      NfcWork work;
      work.data = job->data;
      work.count = (job->status & NFC_JOB_INVALID) ? 0u : job->count;
      work.stride = A.stride;
      work.tiles = A.tiles + job->firstTile;
      A.works[to] = work;
   }
}

/* speculative lanes: all of them (pass 0: every window assumes what the stream's state holds now), or those marked */
/* The waves take the run list in order, and the long lanes of a pass are what its last waves wait for: the lanes go on the
 * list longest first, by classes of their distance to the successor's start (what a lane decodes before it can hand over).
 * order: 0 every lane that runs goes on the list as it comes; 1 all lanes are set up, those of [lenLo, lenHi) go on the
 * list; 2 (launched after 1, class by class) those of [lenLo, lenHi) follow. */
__global__ __launch_bounds__(64) void nfc_window_lanes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, NfcLaunch lanes, uint32_t pass, uint32_t order,
                                                               uint32_t lenLo, uint32_t lenHi)
{
   const uint32_t wi = A.firstWindowSlot + blockIdx.x * blockDim.x + threadIdx.x;
   const uint32_t wEnd = A.firstWindowSlot + (*A.windowCount < A.windowRoom ? *A.windowCount : A.windowRoom);

   if (wi >= wEnd)
      return;

   NfcWindow w = A.windows[wi];
   const NfcScanJob *job = A.jobs + w.job;

   bool listed = true;
   if (order != 0u)
   {
      const uint32_t next = (wi + 1u < wEnd && A.windows[wi + 1u].job == w.job) ? A.windows[wi + 1u].start : job->count;
      const uint32_t len = next - w.start;
      listed = len >= lenLo && len < lenHi;
   }

   if (order == 2u)
   {
      /* (set up by the first launch; `rerun` has been cleared there: the work record says whether the lane runs) */
      if (listed && A.works[wi].count != 0u)
         A.runList[atomicAdd(A.runCount, 1u)] = wi;
      return;
   }

   NfcWork work;
   work.data = job->data + (uint64_t)w.start * A.stride * 4u;
   work.count = 0;
   work.stride = A.stride;
   work.tiles = A.tiles + job->firstTile + w.start / NFC_SCAN_TILE;

   const bool run = !(job->status & NFC_JOB_INVALID) && (pass == 0 || w.rerun);

   if (run)
   {
      if (pass == 0)
         nfc_carry_guess(w.carry, A.windows[w.job].carry, A.points[job->firstPoint + w.start / NFC_SCAN_POINT]);
      else
         w.carry = w.want;

      w.want = w.carry;
      w.rerun = 0;
      w.stop = 0;
      w.retired = 0;
      w.handTo = 0;
      w.pubState = 0;
      w.pubTail = 0;
      w.saved = 0;
      if (pass == 0)
         w.noHand = 0;

      NfcConfig cc;
      nfc_fixed_runtime_config(cfgPtr, cc);

      const uint32_t chunk = job->firstChunk + w.start / A.params.chunkSamples;

      NfcStreamState s;
      NfcStreamCold cold;
      {
         const NfcScanPoint &pt = A.points[job->firstPoint + w.start / NFC_SCAN_POINT];
         w.tracked = (pt.zone & NFC_ZONE_EDGE_KNOWN) ? pt.edgeTime : A.chunkEdge[chunk];
         nfc_window_lane(cc, w, pt, A.states[job->slot].clock, s, cold);
      }

      lanes.states[wi] = s;
      lanes.cold[wi] = cold;
      A.windows[wi] = w;

      work.count = job->count - w.start;

      if (listed)
         A.runList[atomicAdd(A.runCount, 1u)] = wi;
   }

   A.works[wi] = work;
}

/* once the chain is settled: jobs whose last lane is a speculative window get that window set up again in their own
 * final-lane slot (A.finalLaneSlot + job), to be run by the wave decoder (nfc_wave_kernel, mode NFC_WAVE_FINAL) */
__global__ __launch_bounds__(64) void nfc_final_lanes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, NfcLaunch lanes)
{
   const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;

   if (j >= A.nJobs)
      return;

   const NfcScanJob *job = A.jobs + j;
   const uint32_t to = A.finalLaneSlot + j;

   NfcWork work;
   work.data = nullptr;
   work.count = 0;
   work.stride = A.stride;
   work.tiles = nullptr;

   if (!(job->status & NFC_JOB_INVALID) && job->finalLane != j && A.windows[job->finalLane].saved == 0u)
   {
      NfcWindow w = A.windows[job->finalLane];

      NfcConfig cc;
      nfc_fixed_runtime_config(cfgPtr, cc);

      const uint32_t chunk = job->firstChunk + w.start / A.params.chunkSamples;

      NfcStreamState s;
      NfcStreamCold cold;
      {
         const NfcScanPoint &pt = A.points[job->firstPoint + w.start / NFC_SCAN_POINT];
         w.tracked = (pt.zone & NFC_ZONE_EDGE_KNOWN) ? pt.edgeTime : A.chunkEdge[chunk];
         nfc_window_lane(cc, w, pt, A.states[job->slot].clock, s, cold);
      }

      lanes.states[to] = s;
      lanes.cold[to] = cold;
      A.windows[to] = w;

      work.data = job->data + (uint64_t)w.start * A.stride * 4u;
      work.count = job->count - w.start;
      work.tiles = A.tiles + job->firstTile + w.start / NFC_SCAN_TILE;
   }

   A.works[to] = work;
}

/* one thread per job after a decode pass; counts the jobs that need another pass */
__global__ __launch_bounds__(64) void nfc_chain_kernel(NfcScanArgs A, NfcLaunch lanes, uint32_t maxPasses)
{
   const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;

   if (j >= A.nJobs)
      return;

   NfcScanJob job = A.jobs[j];

   if (job.status & NFC_JOB_INVALID)
      return;

   /* settled in an earlier pass: none of its lanes has run since */
   if (job.passes > 0 && !(job.status & NFC_JOB_RERUN))
      return;

   if (nfc_chain_follow(job, j, A.windows, lanes.states, lanes.cold, maxPasses))
      atomicAdd(A.rerunCount, 1u);

   A.jobs[j] = job;
}

/* one wave per job once the chain is settled: frames in stream order into the sink, last lane's state into the stream */
__global__ __launch_bounds__(64) void nfc_finish_kernel(NfcScanArgs A, NfcLaunch real, NfcLaunch lanes)
{
   const uint32_t j = blockIdx.x;
   const uint32_t t = threadIdx.x;

   if (j >= A.nJobs)
      return;

   const NfcScanJob *job = A.jobs + j;

   /* records the staging sink had no room for are lost frames like any other */
   if (j == 0 && t == 0 && lanes.sinkCtl[1])
      atomicAdd(real.sinkCtl + 1, lanes.sinkCtl[1]);

   if (job->status & NFC_JOB_INVALID)
      return;

   if (t == 0)
      nfc_finish_frames(*job, j, A.windows, lanes.cold, lanes.sink, real.sink, real.sinkCtl, real.sinkWords);

   /* the stream's last lane: the carry lane itself, a speculative lane that left its rings in the save area, or one that
    * has been run again in the job's final-lane slot */
   const uint32_t saved = job->finalLane == j ? 0u : A.windows[job->finalLane].saved;
   const uint32_t from = job->finalLane == j ? j : (saved ? job->finalLane : A.finalLaneSlot + j), to = job->slot;
   const uint32_t rows = real.ringBlockFloats / NFC_LANES;

   float *dst = real.rings + (uint64_t)(to / NFC_LANES) * real.ringBlockFloats + (to % NFC_LANES);

   if (saved)
   {
      const float *src = A.saveRings + (uint64_t)(saved - 1u) * rows;

      for (uint32_t i = t; i < rows; i += NFC_LANES)
         dst[(uint64_t)i * NFC_LANES] = src[i];

      for (uint32_t i = t; i < NFC_STREAM_BYTES / 4; i += NFC_LANES)
         ((uint32_t *)(real.bytes + (uint64_t)to * NFC_STREAM_BYTES))[i] = ((const uint32_t *)(A.saveBytes + (uint64_t)(saved - 1u) * NFC_STREAM_BYTES))[i];
   }
   else
   {
      const float *src = lanes.rings + (uint64_t)(from / NFC_LANES) * lanes.ringBlockFloats + (from % NFC_LANES);

      for (uint32_t i = t; i < rows; i += NFC_LANES)
         dst[(uint64_t)i * NFC_LANES] = src[(uint64_t)i * NFC_LANES];

      for (uint32_t i = t; i < NFC_STREAM_BYTES / 4; i += NFC_LANES)
         ((uint32_t *)(real.bytes + (uint64_t)to * NFC_STREAM_BYTES))[i] = ((const uint32_t *)(lanes.bytes + (uint64_t)from * NFC_STREAM_BYTES))[i];
   }

   if (t == 0)
   {
      NfcStreamState s = lanes.states[from];
      NfcStreamCold cold = lanes.cold[from];

      if (job->finalLane != j)
         nfc_final_fixup(s, cold, A.windows[job->finalLane]);

      cold.frameHead = 0;
      cold.frameTail = 0;
      __builtin_memset(cold.boundF, 0, sizeof(cold.boundF)); /* (a lane's notes: nothing of the stream's) */
      __builtin_memset(cold.waitUsed, 0, sizeof(cold.waitUsed));
      cold.waitFlags = 0;
      cold.emitOwn = 0;
      cold.trackedEnd = 0;

      real.states[to] = s;
      real.cold[to] = cold;
   }
}

/* streaming read: 16 bytes per lane per load, four loads in flight, the sums only exist so that the loads are kept */
__global__ __launch_bounds__(256) void nfc_read_kernel(const float4 *__restrict__ data, uint64_t n, float *__restrict__ out)
{
   const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
   uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
   float acc = 0.0f;

   for (; i + 3 * stride < n; i += 4 * stride)
   {
      const float4 a = data[i], b = data[i + stride], c = data[i + 2 * stride], d = data[i + 3 * stride];
      acc += a.x + b.y + c.z + d.w;
   }

   for (; i < n; i += stride)
      acc += data[i].x;

   if (acc == 12345.678f)
      out[0] = acc;
}
