/*
 * TEST INFRASTRUCTURE ONLY - the wave decoder (nfc-laboratory_amd/csrc/nfc_wave.hpp, the text the GPU runs) on the CPU:
 * every wave is 64 fibres (wavesim.hpp), LDS is a struct per wave. Linked into the emulated build of the host runtime
 * (build_emulated.sh) as the twin of nfc_wave_kernel.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "wavesim.hpp"

#define NFC_HIST 1024u
#define NFC_RING_STRIDE 1u
#define NFC_WAVE_LDS
#define NFC_RING_FLOAT float

#define NFC_DEV static inline
static inline uint32_t emu_add(uint32_t *p, uint32_t v) { uint32_t old = *p; *p += v; return old; }
#define NFC_ATOMIC_ADD(ptr, value) emu_add((ptr), (value))
#define NFC_ANY(predicate) (predicate)
#include "../../nfc-laboratory_amd/csrc/nfc_core.hpp"

static inline float emu_sample_at(const uint8_t *data, uint32_t stride, uint32_t i)
{
   const float *p = reinterpret_cast<const float *>(data);
   if (stride == 2)
   {
      volatile float ii = p[2 * i] * p[2 * i];
      volatile float qq = p[2 * i + 1] * p[2 * i + 1];
      return __builtin_sqrtf(ii + qq);
   }
   return p[i];
}
#define NFC_SAMPLE_AT(data, stride, index) emu_sample_at((data), (stride), (index))
#define NFC_FENCE() ((void)0)
#include "../../nfc-laboratory_amd/csrc/nfc_scan.hpp"

#define NFC_WAVE_LANE() (wavesim::lane())
#define NFC_WAVE_BARRIER() wavesim::barrier()
#define NFC_WAVE_BALLOT(p) wavesim::ballot(p)
#define NFC_WAVE_UNIFORM_BEGIN(u) if (wavesim::lane() == 0) {
#define NFC_WAVE_UNIFORM_END(u) } wavesim::uniform_sync(u);
#define NFC_WAVE_STAT_ADD(p, v) (*(p) += (v))
#define NFC_WAVE_STAT_MAX(p, v) (*(p) = *(p) > (v) ? *(p) : (v))

#include "../../nfc-laboratory_amd/csrc/nfc_wave.hpp"

namespace {

struct Call
{
   const NfcConfig *cfg;
   const NfcLaunch *L;
   const NfcScanArgs *A;
   uint32_t mode;
   uint32_t item;
   NfcWaveLds *lds;
};

void lane_body(void *p)
{
   const Call *c = (const Call *)p;
   nfc_wave_run(c->cfg, *c->cfg, *c->L, *c->A, c->mode, c->item, c->lds);
}

} // namespace

void emu_wave_kernel(const NfcConfig *cfgPtr, const NfcLaunch &L, const NfcScanArgs &A, uint32_t mode, uint32_t blocks)
{
   static NfcWaveLds lds;

   for (uint32_t b = 0; b < blocks; b++)
   {
      Call c {cfgPtr, &L, &A, mode, b, &lds};
      wavesim::run(lane_body, &c);
   }
}
