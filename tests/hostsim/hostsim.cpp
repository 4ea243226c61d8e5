/*
 * TEST INFRASTRUCTURE ONLY — CPU build of the device step machine (nfc_core.hpp) so that the
 * state machine can be exercised by `pytest -m "not gpu"` on a box without a GPU.
 * It is never linked into libnfcgpu.so and is not a fallback: the product path is HIP only.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NFC_DEV static inline
static inline uint32_t hostsim_add(uint32_t *p, uint32_t v) { uint32_t old = *p; *p += v; return old; }
#define NFC_ATOMIC_ADD(ptr, value) hostsim_add((ptr), (value))
#define NFC_ANY(predicate) (predicate) /* one stream per call here */
#include "../../nfc-laboratory_amd/csrc/nfc_core.hpp"
#include "../../nfc-laboratory_amd/csrc/nfc_config.hpp"

extern "C" {

struct hostsim_frame
{
   uint32_t stream_id, tech_type, frame_type, frame_flags, frame_phase, frame_rate, length, reserved;
   uint64_t sample_start, sample_end, sample_rate;
   uint8_t data[512];
};

/* decode one stream placed in lane `lane` of a 64-wide stream block; returns frame count */
/* the same after `repeats` copies of an idle buffer (stride 1): takes the 32-bit sample clock to its wrap */
long hostsim_decode_after_idle(const float *idle, uint32_t idleCount, uint64_t repeats, const float *samples, uint64_t count,
                               uint32_t stride, uint32_t sampleRate, uint32_t lane, uint32_t enabled, float powerThreshold,
                               const float *corr, const float *minDepth, const float *maxDepth, hostsim_frame *out, uint32_t cap)
{
   NfcHostParams p;
   p.sampleRate = sampleRate;
   p.enabled = enabled;
   if (powerThreshold == powerThreshold)
      p.powerLevelThreshold = powerThreshold;
   for (int t = 0; t < 4; t++)
   {
      if (corr && corr[t] == corr[t]) p.corrThreshold[t] = corr[t];
      if (minDepth && minDepth[t] == minDepth[t]) p.minDepth[t] = minDepth[t];
      if (maxDepth && maxDepth[t] == maxDepth[t]) p.maxDepth[t] = maxDepth[t];
   }

   NfcConfig cfg;
   if (!nfc_build_config(p, cfg))
      return -1;

   std::vector<float> rings((4 * NFC_HIST + NFC_PROD + cfg.corrTotal) * NFC_LANES, 0.0f);
   std::vector<uint8_t> bytes(NFC_STREAM_BYTES, 0);
   std::vector<uint32_t> arena(1u << 22, 0);

   NfcLaneMem mem;
   mem.ring = rings.data();
   mem.lane = lane;
   mem.exact = true;
   mem.linked = false;
   mem.flags = nullptr;
   mem.bytes = bytes.data();
   uint32_t ctl[2] = {0, 0};
   mem.sink = arena.data();
   mem.sinkCursor = &ctl[0];
   mem.sinkDropped = &ctl[1];
   mem.sinkWords = (uint32_t)arena.size();
   mem.streamId = lane;

   NfcStreamState s;
   NfcStreamCold cold;
   std::memset(&s, 0, sizeof(s));
   std::memset(&cold, 0, sizeof(cold));
   mem.cold = &cold;
   mem.tables = &cfg;
   nfc_state_init(cfg, s, cold, false);

   for (uint64_t r = 0; r < repeats; r++)
   {
      for (uint32_t i = 0; i < idleCount; i++)
         nfc_step(cfg, s, mem, idle[i], nfc_exact_zone(s.clock + 1u));
   }

   for (uint64_t i = 0; i < count; i++)
   {
      float v;
      if (stride == 2)
      {
         volatile float ii = samples[2 * i] * samples[2 * i];
         volatile float qq = samples[2 * i + 1] * samples[2 * i + 1];
         v = __builtin_sqrtf(ii + qq);
      }
      else
         v = samples[i];

      /* same choice as the kernel: exact ring positions near the stream start / clock wrap, incremental otherwise */
      nfc_step(cfg, s, mem, v, nfc_exact_zone(s.clock + 1u));
   }

   long n = 0;
   uint32_t pos = 0;
   if (ctl[1])
      return -2;

   while (pos < ctl[0])
   {
      const uint32_t *w = arena.data() + pos + 1;
      uint32_t len = w[7];
      if (n < cap)
      {
         hostsim_frame &f = out[n];
         std::memset(&f, 0, sizeof(f));
         f.tech_type = w[0]; f.frame_type = w[1]; f.frame_flags = w[2]; f.frame_phase = w[3];
         f.frame_rate = w[4]; f.sample_start = w[5]; f.sample_end = w[6]; f.length = len;
         f.sample_rate = sampleRate;
         std::memcpy(f.data, w + 8, len);
      }
      n++;
      pos += NFC_FRAME_HEADER_WORDS + ((len + 3) >> 2);
   }
   return n;
}

long hostsim_decode(const float *samples, uint64_t count, uint32_t stride, uint32_t sampleRate, uint32_t lane,
                    uint32_t enabled, float powerThreshold, const float *corr, const float *minDepth, const float *maxDepth,
                    hostsim_frame *out, uint32_t cap)
{
   return hostsim_decode_after_idle(nullptr, 0, 0, samples, count, stride, sampleRate, lane, enabled, powerThreshold, corr,
                                    minDepth, maxDepth, out, cap);
}

}
