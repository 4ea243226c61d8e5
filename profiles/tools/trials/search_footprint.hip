/* Static trial (never linked into the product, not run): register and instruction footprint of search-only kernels
 * restricted to a subset of the detector bank, under occupancy constraints. Answers "how many waves per SIMD would a
 * wave specialised on one technology's detectors fit?" for DESIGN.md section 8.
 *   cd profiles/tools/trials && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -simplifycfg-sink-common=false \
 *      -I../../../nfc-laboratory_amd/csrc -I../../../nfc-laboratory_amd/build -I../../../include -S --offload-device-only \
 *      search_footprint.hip -o /tmp/search_footprint.s      then read .amdhsa_next_free_vgpr / private_segment_fixed_size
 * Round-1 result (registers, scratch bytes): all detectors 256/296 at 2 waves, 168/724 at 3; NFC-A alone 166/72 at 3,
 * 128/204 at 4; NFC-B alone 112/0 at 4; NFC-F alone 128/144 at 4; NFC-V alone 100/0 at 4, 64/124 at 8; front end alone 47/0.
 * Note the LDS tile: 64 x 65 floats per wave caps a CU at 9 waves whatever the registers; the trial uses 64 x 17. */
#include <hip/hip_runtime.h>
#include <stdint.h>
#define NFC_DEV __device__ __forceinline__
#define NFC_ATOMIC_ADD(ptr, value) atomicAdd((ptr), (value))
#define NFC_ANY(predicate) (__any(predicate) != 0)
#include "nfc_core.hpp"
#include "nfc_launch.h"
#define NFC_FIXED_FN __device__ __forceinline__
#include "nfc_config_fixed.inc"

// search-only step: front end + the detector subset selected by MASK, no decode path at all
template <uint32_t MASK>
__device__ __forceinline__ void search_step(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float value)
{
   ++s.clock; ++s.pulseFilter;
   nfc_advance_positions(c, s, mem);
   NfcTapsA ta; NfcTapsB tb; NfcTapsF tf; NfcTapsV tv;
   if (MASK & 1) nfca_load_taps(c, s, mem, ta);
   if (MASK & 2) nfcb_load_taps(c, s, mem, tb);
   if (MASK & 4) nfcf_load_taps(c, s, mem, tf);
   if (MASK & 8) nfcv_load_taps(c, s, mem, tv);
   const NfcNow now = nfc_front_end(c, s, mem, value);
   nfc_detect_carrier(c, s, mem);
   const bool armed = s.clock >= 1024u && !(s.env < c.powerThreshold);
   if (armed)
   {
      uint32_t locked = 0;
      if ((MASK & 1) && nfca_detect(c, s, mem, ta, now)) locked = NFC_TECH_A;
      else if ((MASK & 2) && nfcb_detect(c, s, mem, tb, now)) locked = NFC_TECH_B;
      else if ((MASK & 4) && nfcf_detect(c, s, mem, tf, now)) locked = NFC_TECH_F;
      else if ((MASK & 8) && nfcv_detect(c, s, mem, tv, now)) locked = NFC_TECH_V;
      if (locked) s.lockTech = locked;   // the hand-over itself is not part of the trial
   }
}

template <uint32_t MASK>
__device__ __forceinline__ void body(const NfcConfig *cfgPtr, NfcLaunch L, float *tile)
{
   const uint32_t lane = threadIdx.x, block = L.firstBlock + blockIdx.x, slot = block * NFC_LANES + lane;
   NfcStreamState s;
   __builtin_memset(&s, 0, sizeof(s));
   const NfcStreamState &g = L.states[slot];
   s.clock = g.clock; s.pulseFilter = g.pulseFilter; s.env = g.env; s.n1 = g.n1; s.mdev = g.mdev; s.avg = g.avg; s.edgePeak = g.edgePeak;
   s.edgeTime = g.edgeTime; s.carrierOff = g.carrierOff; s.carrierOn = g.carrierOn; s.bankClock = g.bankClock;
   if (MASK & 1) { for (int r = 0; r < 3; r++) { s.posA[r] = g.posA[r]; s.u.search.detA[r] = g.u.search.detA[r]; } }
   if (MASK & 2) { s.u.search.detB[0] = g.u.search.detB[0]; s.u.search.detB[1] = g.u.search.detB[1]; }
   if (MASK & 4) { for (int r = 0; r < 2; r++) { s.posF[r] = g.posF[r]; s.u.search.detF[r] = g.u.search.detF[r]; } }
   if (MASK & 8) { s.posV1 = g.posV1; s.posV0 = g.posV0; s.u.search.detV = g.u.search.detV; }
   NfcLaneMem mem;
   mem.linked = false;
   mem.flags = nullptr;
   mem.ring = L.rings + (uint64_t)block * L.ringBlockFloats; mem.lane = lane; mem.exact = false;
   mem.bytes = L.bytes + (uint64_t)slot * NFC_STREAM_BYTES; mem.sink = L.sink; mem.sinkCursor = L.sinkCtl;
   mem.sinkDropped = L.sinkCtl + 1; mem.sinkWords = L.sinkWords; mem.streamId = slot; mem.cold = L.cold + slot; mem.tables = cfgPtr;
   NfcConfig cc; nfc_fixed_config(cc);
   cc.enabled = cfgPtr->enabled; cc.powerThreshold = cfgPtr->powerThreshold; cc.lowThreshold = cfgPtr->lowThreshold; cc.highThreshold = cfgPtr->highThreshold;
   for (int t = 0; t < 4; t++) { cc.corrThreshold[t] = cfgPtr->corrThreshold[t]; cc.minDepth[t] = cfgPtr->minDepth[t]; cc.maxDepth[t] = cfgPtr->maxDepth[t]; }
   for (uint32_t k = 0; k < L.uniformCount; k++)
      search_step<MASK>(cc, s, mem, tile[lane * 17 + (k & 15)]);
   NfcStreamState &o = L.states[slot];
   o.clock = s.clock; o.pulseFilter = s.pulseFilter; o.env = s.env; o.n1 = s.n1; o.mdev = s.mdev; o.avg = s.avg; o.edgePeak = s.edgePeak;
   o.edgeTime = s.edgeTime; o.carrierOff = s.carrierOff; o.carrierOn = s.carrierOn; o.bankClock = s.bankClock; o.lockTech = s.lockTech;
   if (MASK & 1) { for (int r = 0; r < 3; r++) { o.posA[r] = s.posA[r]; o.u.search.detA[r] = s.u.search.detA[r]; } }
   if (MASK & 2) { o.u.search.detB[0] = s.u.search.detB[0]; o.u.search.detB[1] = s.u.search.detB[1]; }
   if (MASK & 4) { for (int r = 0; r < 2; r++) { o.posF[r] = s.posF[r]; o.u.search.detF[r] = s.u.search.detF[r]; } }
   if (MASK & 8) { o.posV1 = s.posV1; o.posV0 = s.posV0; o.u.search.detV = s.u.search.detV; }
}
#define K(name, mask, w) __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(w, w))) void name(const NfcConfig *cfgPtr, NfcLaunch L) { __shared__ float tile[64*17]; tile[threadIdx.x] = L.rings[threadIdx.x]; __syncthreads(); body<mask>(cfgPtr, L, tile); }
K(trial_all_w2, 15, 2)
K(trial_all_w3, 15, 3)
K(trial_a_w3, 1, 3)
K(trial_a_w4, 1, 4)
K(trial_a_w5, 1, 5)
K(trial_b_w4, 2, 4)
K(trial_b_w6, 2, 6)
K(trial_b_w8, 2, 8)
K(trial_f_w4, 4, 4)
K(trial_f_w5, 4, 5)
K(trial_f_w6, 4, 6)
K(trial_v_w4, 8, 4)
K(trial_v_w6, 8, 6)
K(trial_v_w8, 8, 8)
K(trial_bfv_w3, 14, 3)
K(trial_bfv_w4, 14, 4)
K(trial_bv_w4, 10, 4)
K(trial_bv_w5, 10, 5)
K(trial_none_w8, 0, 8)
