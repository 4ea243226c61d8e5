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
#define NFC_WAVE_BARRIER() __syncthreads()
#define NFC_WAVE_BALLOT(p) ((uint64_t)__ballot(p))
#define NFC_WAVE_UNIFORM_BEGIN {
#define NFC_WAVE_UNIFORM_END }
#define NFC_WAVE_UNIFORM_U32(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
/* a word a uniform block hands to the code after it: every lane has computed it (no trip through LDS) */
#define NFC_WAVE_UNIFORM_LEAVE(slot, value) ((void)0)
#define NFC_WAVE_UNIFORM_TAKE(slot, value) ((value) = (uint32_t)__builtin_amdgcn_readfirstlane((int)(value)))
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

#include "nfc_wave.hpp"

__global__ __launch_bounds__(64) NFC_WAVE_KERNEL_ATTR void nfc_wave_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, NfcScanArgs A, uint32_t mode)
{
   __shared__ NfcWaveLds lds;
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
