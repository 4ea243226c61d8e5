/* Static trial (never linked into the product, not run): register footprint of a tile-major search kernel - front end
 * over a tile first (per-sample results parked in LDS), then one technology's detectors at a time with that technology's
 * records loaded from / stored to memory around its pass. No lock hand-over, take-back or decode mode: this only answers
 * how many registers the passes need when the other technologies' records are not live.
 *   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -simplifycfg-sink-common=false \
 *      -I../../../nfc-laboratory_amd/csrc -I../../../nfc-laboratory_amd/build -I../../../include -S --offload-device-only \
 *      tilemajor_footprint.hip -o /tmp/tilemajor_footprint.s
 * Round-1 result, tile of 16 samples, 8.7 KB of LDS per wave (registers / scratch bytes / scratch instructions in the
 * whole kernel): 214 / 0 / 0 unconstrained, 168 / 140 / 20 at three waves per SIMD, 128 / 292 / 50 at four, 96 / 416 / 109
 * at five. The per-sample search kernel of the same detectors (search_footprint.hip) needs 256 / 296 / 85 at two waves
 * and 168 / 724 / 204 at three. NFC-A (three rates in one pass) is the largest pass. */
#include <hip/hip_runtime.h>
#include <stdint.h>
#define NFC_DEV __device__ __forceinline__
#define NFC_ATOMIC_ADD(ptr, value) atomicAdd((ptr), (value))
#define NFC_ANY(predicate) (__any(predicate) != 0)
#include "nfc_core.hpp"
#include "nfc_launch.h"
#define NFC_FIXED_FN __device__ __forceinline__
#include "nfc_config_fixed.inc"

#ifndef T
#define T 16
#endif

struct Parked /* one block's detector records, technology by technology (a real kernel would lay them out [word][lane]) */
{
   NfcDetA a[3];
   NfcDetB b[2];
   NfcDetF f[2];
   NfcDetV v;
};

template <int W>
__device__ __forceinline__ void body(const NfcConfig *cfgPtr, NfcLaunch L, Parked *parked, float *lds)
{
   const uint32_t lane = threadIdx.x, block = L.firstBlock + blockIdx.x, slot = block * NFC_LANES + lane;
   float *tileIn = lds, *tileE = lds + 64 * (T + 1); /* the other per-sample results of the front end are read back from the history rings */

   NfcLaneMem mem;
   mem.linked = false;
   mem.flags = nullptr;
   mem.ring = L.rings + (uint64_t)block * L.ringBlockFloats; mem.lane = lane; mem.exact = false;
   mem.bytes = L.bytes + (uint64_t)slot * NFC_STREAM_BYTES; mem.sink = L.sink; mem.sinkCursor = L.sinkCtl;
   mem.sinkDropped = L.sinkCtl + 1; mem.sinkWords = L.sinkWords; mem.streamId = slot; mem.cold = L.cold + slot; mem.tables = cfgPtr;
   NfcConfig cc; nfc_fixed_config(cc);
   cc.enabled = cfgPtr->enabled; cc.powerThreshold = cfgPtr->powerThreshold; cc.lowThreshold = cfgPtr->lowThreshold; cc.highThreshold = cfgPtr->highThreshold;
   for (int t = 0; t < 4; t++) { cc.corrThreshold[t] = cfgPtr->corrThreshold[t]; cc.minDepth[t] = cfgPtr->minDepth[t]; cc.maxDepth[t] = cfgPtr->maxDepth[t]; }

   NfcStreamState s;
   __builtin_memset(&s, 0, sizeof(s));
   const NfcStreamState &g = L.states[slot];
   s.clock = g.clock; s.pulseFilter = g.pulseFilter; s.env = g.env; s.n1 = g.n1; s.mdev = g.mdev; s.avg = g.avg; s.edgePeak = g.edgePeak;
   s.edgeTime = g.edgeTime; s.carrierOff = g.carrierOff; s.carrierOn = g.carrierOn; s.bankClock = g.bankClock;
   for (int r = 0; r < 3; r++) s.posA[r] = g.posA[r];
   for (int r = 0; r < 2; r++) s.posF[r] = g.posF[r];
   s.posV1 = g.posV1; s.posV0 = g.posV0;

   Parked &pk = parked[slot];
   uint32_t lockAt = T, lockTech = 0;

   for (uint32_t base = 0; base < L.uniformCount; base += T)
   {
      const uint32_t clock0 = s.clock;

      /* front end + carrier over the tile */
      for (uint32_t k = 0; k < T; k++)
      {
         ++s.clock; ++s.pulseFilter;
         const NfcNow now = nfc_front_end(cc, s, mem, tileIn[lane * (T + 1) + k]);
         nfc_detect_carrier(cc, s, mem);
         tileE[lane * (T + 1) + k] = s.env;
      }
      const uint32_t clockEnd = s.clock;
      const float envEnd = s.env;

#define PASS(LOAD, STORE, TAPS, DETECT, TECH)                                                            \
      {                                                                                                  \
         LOAD;                                                                                           \
         for (uint32_t k = 0; k < lockAt; k++)                                                           \
         {                                                                                               \
            s.clock = clock0 + 1 + k; s.env = tileE[lane * (T + 1) + k];                                 \
            NfcNow now; now.x = NFC_AT(mem, NFC_R_X, s.clock & NFC_HMASK); now.filt = NFC_AT(mem, NFC_R_FILT, s.clock & NFC_HMASK); \
            now.mdev = NFC_AT(mem, NFC_R_MDEV, s.clock & NFC_HMASK); now.depth = NFC_AT(mem, NFC_R_DEPTH, s.clock & NFC_HMASK); \
            if (s.clock >= 1024u && !(s.env < cc.powerThreshold))                                        \
            {                                                                                            \
               TAPS;                                                                                     \
               if (DETECT) { lockAt = k; lockTech = TECH; }                                              \
            }                                                                                            \
         }                                                                                               \
         STORE;                                                                                          \
      }

      PASS(for (int r = 0; r < 3; r++) s.u.search.detA[r] = pk.a[r], for (int r = 0; r < 3; r++) pk.a[r] = s.u.search.detA[r],
           mem.exact = true; nfc_advance_positions(cc, s, mem); NfcTapsA ta; nfca_load_taps(cc, s, mem, ta), nfca_detect(cc, s, mem, ta, now), NFC_TECH_A)
      PASS(for (int r = 0; r < 2; r++) s.u.search.detB[r] = pk.b[r], for (int r = 0; r < 2; r++) pk.b[r] = s.u.search.detB[r],
           NfcTapsB tb; nfcb_load_taps(cc, s, mem, tb), nfcb_detect(cc, s, mem, tb, now), NFC_TECH_B)
      PASS(for (int r = 0; r < 2; r++) s.u.search.detF[r] = pk.f[r], for (int r = 0; r < 2; r++) pk.f[r] = s.u.search.detF[r],
           mem.exact = true; nfc_advance_positions(cc, s, mem); NfcTapsF tf; nfcf_load_taps(cc, s, mem, tf), nfcf_detect(cc, s, mem, tf, now), NFC_TECH_F)
      PASS(s.u.search.detV = pk.v, pk.v = s.u.search.detV,
           mem.exact = true; nfc_advance_positions(cc, s, mem); NfcTapsV tv; nfcv_load_taps(cc, s, mem, tv), nfcv_detect(cc, s, mem, tv, now), NFC_TECH_V)

      s.clock = clockEnd; s.env = envEnd;
      lockAt = T;
   }

   NfcStreamState &o = L.states[slot];
   o.clock = s.clock; o.pulseFilter = s.pulseFilter; o.env = s.env; o.n1 = s.n1; o.mdev = s.mdev; o.avg = s.avg; o.edgePeak = s.edgePeak;
   o.edgeTime = s.edgeTime; o.carrierOff = s.carrierOff; o.carrierOn = s.carrierOn; o.bankClock = s.bankClock; o.lockTech = lockTech;
}

#define K(name, w) __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(w, w))) void name(const NfcConfig *cfgPtr, NfcLaunch L, Parked *parked) \
   { __shared__ float lds[2 * 64 * (T + 1)]; lds[threadIdx.x] = L.rings[threadIdx.x]; __syncthreads(); body<w>(cfgPtr, L, parked, lds); }
K(tile_w2, 2)
K(tile_w3, 3)
K(tile_w4, 4)
K(tile_w5, 5)
K(tile_w6, 6)
