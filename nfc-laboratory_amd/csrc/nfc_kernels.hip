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
#include "nfc_launch.h"

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

/* magnitude of one IQ sample, the reference's scalar formula (RadioDeviceTask.cpp:626-642): products and sum rounded
 * separately (no contraction), correctly rounded square root */
__device__ __forceinline__ float nfc_iq_magnitude(float i, float q)
{
   return __builtin_sqrtf(__fadd_rn(__fmul_rn(i, i), __fmul_rn(q, q)));
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
