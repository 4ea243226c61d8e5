/*
 * nfc_wave.hip — the wave decoder's kernel (nfc_wave.hpp) for gfx950: a workgroup is one wave, a wave is one lane of
 * work of the time-parallel path. The decoder's step machine is compiled here a second time with one stream's rings in
 * LDS (history 1024 deep, no lane pitch).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

/* the raw history is as deep as the stored one; the samples a tile written ahead displaces are kept beside it
 * (nfc_wave.hpp). The histories of the filtered signal, its deviation and the modulation depth are kept 256 deep: what
 * NFC-V reads further back comes from the front end's planes (nfc_wave_f_deep) */
#define NFC_X_OLD_INDEX(mem, sampleClock) nfc_wave_x_old_index((mem).ring, (sampleClock))
#define NFC_RING_STRIDE 1u
#define NFC_WAVE_LDS __attribute__((address_space(3)))
#define NFC_RING_FLOAT NFC_WAVE_LDS float
#define NFC_HIST_F 256u
struct NfcWaveDeep;
#define NFC_F_DEEP_CTX const NFC_WAVE_LDS NfcWaveDeep *
#define NFC_F_DEEP(mem, region, clk) nfc_wave_f_deep((mem), (region), (clk))
template <class Mem>
__device__ __forceinline__ float nfc_wave_f_deep(const Mem &mem, uint32_t region, uint32_t clk);

#define NFC_DEV __device__ __forceinline__

/* the step runs wave-uniform: one lane appends to the frame sink for all */
__device__ __forceinline__ uint32_t nfc_wave_atomic_add(uint32_t *p, uint32_t v)
{
   uint32_t old = 0;
   if (threadIdx.x == 0)
      old = atomicAdd(p, v);
   return (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
}

#define NFC_ATOMIC_ADD(ptr, value) nfc_wave_atomic_add((ptr), (value))
#define NFC_ANY(predicate) (predicate)

#include "nfc_types.h"

/* index, from the start of the wave's ring storage, of the raw sample of clock `clk`: in the history, or among the samples
 * the tile written ahead has displaced (layout: nfc_wave.hpp, NFC_WAVE_XOLD) */
__device__ __forceinline__ uint32_t nfc_wave_x_old_index(const NFC_RING_FLOAT *ring, uint32_t clk)
{
   const uint32_t displaced = NFC_HIST + 3u * NFC_HIST_F + NFC_PROD + NFC_CORR_MAX;
   const uint32_t clock0 = __builtin_bit_cast(uint32_t, (float)ring[displaced + NFC_LANES]);
   const uint32_t k = clk - (clock0 - (NFC_HIST - 1u)); /* sample clock0 - 511 + j was displaced by tile sample j */
   return k < NFC_LANES ? displaced + k : (clk & (NFC_HIST - 1u));
}

#include "nfc_core.hpp"

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
#include "nfc_scan.hpp"

#define NFC_FIXED_FN __device__ __forceinline__
#include "nfc_config_fixed.inc"

/* run-time part of the configuration on top of the compiled-in table (the path is only taken at that sample rate) */
__device__ __forceinline__ void nfc_wave_config(const NfcConfig *cfgPtr, NfcConfig &cc)
{
   typedef __attribute__((address_space(4))) const NfcConfig ConstConfig;
   ConstConfig *cp = (ConstConfig *)cfgPtr;

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

/* the same inside the step functions: the run-time part from where nfc_wave_run parked it in LDS (NfcWaveLds::cfg) */
__device__ __forceinline__ void nfc_wave_config_parked(const NFC_WAVE_LDS uint32_t *parked, NfcConfig &cc)
{
   nfc_fixed_config(cc);
   cc.enabled = parked[0];
   cc.powerThreshold = __builtin_bit_cast(float, parked[1]);
   cc.lowThreshold = __builtin_bit_cast(float, parked[2]);
   cc.highThreshold = __builtin_bit_cast(float, parked[3]);
   for (int t = 0; t < 4; t++)
   {
      cc.corrThreshold[t] = __builtin_bit_cast(float, parked[4 + t]);
      cc.minDepth[t] = __builtin_bit_cast(float, parked[8 + t]);
      cc.maxDepth[t] = __builtin_bit_cast(float, parked[12 + t]);
   }
}

/* wave-wide inclusive prefix sum and maximum on the data-parallel primitives of the SIMD: shifts inside the rows of 16
 * lanes, then the row totals handed on (row_bcast:15, row_bcast:31) */
#define NFC_DPP_F(old, src, ctrl, rows) \
   __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(src)), (ctrl), (rows), 0xf, false))

__device__ __forceinline__ float nfc_wave_scan_add(float v)
{
   v += NFC_DPP_F(0.0f, v, 0x111, 0xf); /* row_shr:1 */
   v += NFC_DPP_F(0.0f, v, 0x112, 0xf); /* row_shr:2 */
   v += NFC_DPP_F(0.0f, v, 0x114, 0xf); /* row_shr:4 */
   v += NFC_DPP_F(0.0f, v, 0x118, 0xf); /* row_shr:8 */
   v += NFC_DPP_F(0.0f, v, 0x142, 0xa); /* row_bcast:15 into rows 1 and 3 */
   v += NFC_DPP_F(0.0f, v, 0x143, 0xc); /* row_bcast:31 into rows 2 and 3 */
   return v;
}

__device__ __forceinline__ float nfc_wave_max(float v)
{
   float t;
   t = NFC_DPP_F(v, v, 0x111, 0xf); v = t > v ? t : v;
   t = NFC_DPP_F(v, v, 0x112, 0xf); v = t > v ? t : v;
   t = NFC_DPP_F(v, v, 0x114, 0xf); v = t > v ? t : v;
   t = NFC_DPP_F(v, v, 0x118, 0xf); v = t > v ? t : v;
   t = NFC_DPP_F(v, v, 0x142, 0xa); v = t > v ? t : v;
   t = NFC_DPP_F(v, v, 0x143, 0xc); v = t > v ? t : v;
   return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

#define NFC_WAVE_LANE() (threadIdx.x)
/* A workgroup is one wavefront, and a wavefront's LDS instructions are carried out in the order they were issued: what one lane
 * has written another lane reads after it without waiting for anything. __syncthreads() would also wait for every global
 * access in flight - the next tile's samples and planes, fetched ahead on purpose - and for the LDS queue to drain, at each of
 * the dozen points per tile where the lanes hand values to each other. What is needed there is that the compiler keeps the
 * order of the accesses: a fence at wavefront scope. (-DNFC_WAVE_HARD_BARRIER: the workgroup barrier, as until round 4) */
#ifdef NFC_WAVE_HARD_BARRIER
#define NFC_WAVE_BARRIER() __syncthreads()
#else
#define NFC_WAVE_BARRIER()                                          \
   do                                                               \
   {                                                                \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
      __builtin_amdgcn_wave_barrier();                              \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        \
   } while (0)
#endif
#define NFC_WAVE_BALLOT(p) ((uint64_t)__ballot(p))
#define NFC_WAVE_UNIFORM_BEGIN {
#define NFC_WAVE_UNIFORM_END }
#define NFC_WAVE_UNIFORM_U32(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
/* a word a uniform block hands to the code after it: every lane has computed it (no trip through LDS) */
#define NFC_WAVE_UNIFORM_LEAVE(slot, value) ((void)0)
#define NFC_WAVE_UNIFORM_TAKE(slot, value) ((value) = (uint32_t)__builtin_amdgcn_readfirstlane((int)(value)))
#define NFC_WAVE_UNIFORM_TAKE_BITS(slot, value) ((value) = (uint32_t)__builtin_amdgcn_readfirstlane((int)(value))) /* (the slot is a float) */
#define NFC_WAVE_SCAN_ADD_F(v) nfc_wave_scan_add(v)
#define NFC_WAVE_MAX_F(v) nfc_wave_max(v)
#define NFC_WAVE_PICK_F(reg, array, j) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)(reg)), (int)(j)))
#define NFC_WAVE_PICK_U32(reg, array, j) ((uint32_t)__builtin_amdgcn_readlane((int)(reg), (int)(j)))
#define NFC_WAVE_SHFL_F(reg, j) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)(reg)), (int)(j)))
#define NFC_WAVE_CONFIG(cfgPtr, lds, cc) nfc_wave_config_parked((lds)->cfg, (cc))
/* The two step functions are inlined into the tile loop and work on the decoder state where it lies in LDS (round 4: as
 * functions of their own that took the 91-word state into registers they used every register of the wave, and the 66
 * callee-saved ones went to scratch and back on every call: two thirds of the kernel's HBM writes).
 * Experiments: -DNFC_WAVE_STEP_CALL (functions of their own), -DNFC_WAVE_STEP_COPY (state into registers and back),
 * -DNFC_WAVE_WAVES=n (register budget for n waves per SIMD) */
#ifndef NFC_WAVE_STEP_CALL
#define NFC_WAVE_NOINLINE __device__ __forceinline__
#else
#define NFC_WAVE_NOINLINE static __device__ __attribute__((noinline))
#endif
#ifdef NFC_WAVE_WAVES
#define NFC_WAVE_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(NFC_WAVE_WAVES, NFC_WAVE_WAVES)))
#else
#define NFC_WAVE_KERNEL_ATTR
#endif
#define NFC_WAVE_STAT_ADD(p, v) atomicAdd((p), (v))
#define NFC_WAVE_STAT_MAX(p, v) atomicMax((p), (v))

#ifdef NFC_WAVE_VERIFY
/* -DNFC_WAVE_VERIFY (not a product build; one run per round: profiles/tools/r05/wave_verify_device.py): every tile is decoded
 * twice on the device - with the bulk paths of nfc_wave_fast.hpp and again sample by sample by the step machine alone - and
 * everything the two leave is compared word for word: decoder state, protocol state, history and correlation rings, frame
 * bytes, usage marks (what tests/hostsim/emu_wave.cpp does on the CPU build with NFC_EMU_WAVE_VERIFY=1). Tiles verified and
 * tiles that differ are counted in the submission's counter block (words 12 / 13; 14 / 15: where the first difference was). */
#include "nfc_scan_launch.h"
struct NfcWaveLds;
struct NfcWaveItem;
struct NfcWaveSink;
struct NfcWaveFetch;
__device__ void nfc_wave_verify_tile(const NfcConfig *cfgPtr, const NfcConfig &cc, const NfcScanArgs &A, const NfcWaveItem &it, NFC_WAVE_LDS NfcWaveLds *lds,
                                     const NfcWaveSink &sink, uint32_t n, uint32_t pos, bool carry, uint32_t warmFront, uint32_t warm, const NfcWaveFetch &fetched);
#define NFC_WAVE_TILE_HOOK nfc_wave_verify_tile
#endif

#include "nfc_wave.hpp"

#ifdef NFC_WAVE_VERIFY
__device__ uint32_t *nfcWaveVerifyCounters; /* (set by the kernel from its launch record: NfcLaunch::laneStats + 8) */

__device__ __forceinline__ void nfc_wave_verify_copy(NFC_WAVE_LDS NfcWaveLds *to, const NFC_WAVE_LDS NfcWaveLds *from)
{
   NFC_WAVE_LDS uint32_t *t = (NFC_WAVE_LDS uint32_t *)to;
   const NFC_WAVE_LDS uint32_t *f = (const NFC_WAVE_LDS uint32_t *)from;

   for (uint32_t i = threadIdx.x; i < sizeof(NfcWaveLds) / 4u; i += NFC_LANES)
      t[i] = f[i];
}

__device__ void nfc_wave_verify_tile(const NfcConfig *cfgPtr, const NfcConfig &cc, const NfcScanArgs &A, const NfcWaveItem &it, NFC_WAVE_LDS NfcWaveLds *lds,
                                     const NfcWaveSink &sink, uint32_t n, uint32_t pos, bool carry, uint32_t warmFront, uint32_t warm, const NfcWaveFetch &fetched)
{
   __shared__ uint32_t dummy[NFC_FRAME_MAX_WORDS + 16];
   __shared__ uint32_t dummyCtl[2];

   NFC_WAVE_LDS NfcWaveLds *saved = lds + 1, *fast = lds + 2;
   const uint32_t lane = threadIdx.x;

   __syncthreads();
   nfc_wave_verify_copy(saved, lds);
   __syncthreads();

   nfc_wave_tile(cfgPtr, cc, A, it, lds, sink, n, pos, carry, warmFront, warm, fetched, true);

   __syncthreads();
   nfc_wave_verify_copy(fast, lds);
   __syncthreads();
   nfc_wave_verify_copy(lds, saved);
   __syncthreads();

   if (lane == 0)
   {
      /* (the lane's frame records are chained by their place in the sink: the stepped run's go to a sink of its own) */
      lds->cold.frameHead = 0;
      lds->cold.frameTail = 0;
      dummyCtl[0] = 0;
      dummyCtl[1] = 0;
   }
   __syncthreads();

   NfcWaveSink quiet = sink;
   quiet.words = (uint32_t *)dummy;
   quiet.ctl = (uint32_t *)dummyCtl;
   quiet.capacity = NFC_FRAME_MAX_WORDS + 16;

   nfc_wave_tile(cfgPtr, cc, A, it, lds, quiet, n, pos, carry, warmFront, warm, fetched, false);

   __syncthreads();

   /* word for word: state, where the tile loop stands, protocol state (but the chain of frame records), rings (but the product
    * ring, written ahead by the bulk paths), frame bytes, usage marks */
   bool differs = false;
   {
      const NFC_WAVE_LDS uint32_t *a = (const NFC_WAVE_LDS uint32_t *)&fast->u.s, *b = (const NFC_WAVE_LDS uint32_t *)&lds->u.s;
      for (uint32_t i = lane; i < sizeof(NfcStreamState) / 4u; i += NFC_LANES)
         differs = differs || a[i] != b[i];
   }
   {
      const NFC_WAVE_LDS uint32_t *a = (const NFC_WAVE_LDS uint32_t *)&fast->cold, *b = (const NFC_WAVE_LDS uint32_t *)&lds->cold;
      const uint32_t head = (uint32_t)offsetof(NfcStreamCold, frameHead) / 4u, tail = (uint32_t)offsetof(NfcStreamCold, frameTail) / 4u;
      for (uint32_t i = lane; i < sizeof(NfcStreamCold) / 4u; i += NFC_LANES)
         differs = differs || (i != head && i != tail && a[i] != b[i]);
   }
   {
      const NFC_WAVE_LDS uint32_t *a = (const NFC_WAVE_LDS uint32_t *)fast->ring, *b = (const NFC_WAVE_LDS uint32_t *)lds->ring;
      for (uint32_t i = lane; i < NFC_R_PROD; i += NFC_LANES)
         differs = differs || a[i] != b[i];
      for (uint32_t i = lane; i < NFC_CORR_MAX; i += NFC_LANES)
         differs = differs || a[NFC_R_CORR + i] != b[NFC_R_CORR + i];
   }
   {
      const NFC_WAVE_LDS uint32_t *a = (const NFC_WAVE_LDS uint32_t *)fast->bytes, *b = (const NFC_WAVE_LDS uint32_t *)lds->bytes;
      for (uint32_t i = lane; i < NFC_STREAM_BYTES / 4u; i += NFC_LANES)
         differs = differs || a[i] != b[i];
   }
   differs = differs || fast->flags != lds->flags || fast->u.at != lds->u.at;

   const bool any = __ballot(differs) != 0ull;

   if (lane == 0 && nfcWaveVerifyCounters)
   {
      atomicAdd(nfcWaveVerifyCounters + 0, 1u);
      if (any && atomicAdd(nfcWaveVerifyCounters + 1, 1u) == 0u)
      {
         nfcWaveVerifyCounters[2] = pos;
         nfcWaveVerifyCounters[3] = it.w;
      }
   }

   /* go on from the first run (its frames are the ones in the sink) */
   __syncthreads();
   nfc_wave_verify_copy(lds, fast);
   __syncthreads();
}
#endif

/* (experiments compile this text a second time under another name: NFC_WAVE_KERNEL_NAME, e.g. profiles/r06/NOTES.md "a build
 * for few lanes") */
#ifndef NFC_WAVE_KERNEL_NAME
#define NFC_WAVE_KERNEL_NAME nfc_wave_kernel
#endif

__global__ __launch_bounds__(64) NFC_WAVE_KERNEL_ATTR void NFC_WAVE_KERNEL_NAME(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, NfcScanArgs A, uint32_t mode)
{
#ifdef NFC_WAVE_VERIFY
   __shared__ NfcWaveLds ldsThree[3]; /* the wave's, a copy of it as the tile found it, and what the bulk paths left */
#define lds ldsThree[0]
   if (threadIdx.x == 0 && blockIdx.x == 0)
      nfcWaveVerifyCounters = L.laneStats + 8;
#else
   __shared__ NfcWaveLds lds;
#endif
#ifdef NFC_WAVE_LDS_PAD
   /* experiment: more LDS per wave = fewer waves per CU (how much of the throughput is occupancy?) */
   __shared__ uint32_t pad[NFC_WAVE_LDS_PAD / 4];
   pad[(threadIdx.x * 97u + mode) % (NFC_WAVE_LDS_PAD / 4)] = blockIdx.x;
   __syncthreads();
   if (pad[(blockIdx.x * 31u) % (NFC_WAVE_LDS_PAD / 4)] == 0xFFFFFFFFu)
      return;
#endif

   NfcConfig cc;
   nfc_wave_config(cfgPtr, cc);

   nfc_wave_run(cfgPtr, cc, L, A, mode, blockIdx.x, (NFC_WAVE_LDS NfcWaveLds *)&lds);
}
