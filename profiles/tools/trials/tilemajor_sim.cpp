/*
 * DESIGN PROTOTYPE, CPU only, test infrastructure (never linked into the product).
 *
 * Question: can the per-sample order of nfc_step_as() (front end, then the whole detector bank, sample by sample) be
 * replaced by a tile-major order - the front end over a tile of samples first, then one technology's detectors over the
 * tile, then the next technology's - without changing a single frame? That order is what a kernel needs whose waves
 * keep one technology's detector records in registers at a time. The catch is the lock: the first
 * detector to recognise a start of frame freezes all the others from the next sample on, so passes that ran past that
 * sample must be taken back.
 *
 * This file runs the product's device functions (nfc_core.hpp, unchanged) for one stream in that order:
 *   decode mode            sample by sample, as today (nfc_step);
 *   search mode, per tile  (a) front end + carrier detection over the rest of the tile, speculatively, remembering the
 *                              scalars after every sample and what was emitted;
 *                          (b) NFC-A, B, F, V in bank order, each over the samples before the earliest lock found so
 *                              far (the sample of the lock included for the technologies ahead of the locking one in
 *                              bank order); a later technology locking earlier takes the earlier ones back to that
 *                              sample (restore the records of the tile start, run again);
 *                          (c) on a lock: front-end scalars and emitted frames taken back to the lock sample, decode
 *                              mode entered, the rest of the tile continues sample by sample.
 * profiles/tools/trials/tilemajor_check.py compares the frames with the reference decoder.
 */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NFC_DEV static inline
static inline uint32_t sim_add(uint32_t *p, uint32_t v) { uint32_t old = *p; *p += v; return old; }
#define NFC_ATOMIC_ADD(ptr, value) sim_add((ptr), (value))
#define NFC_ANY(predicate) (predicate)
#include "../../../nfc-laboratory_amd/csrc/nfc_core.hpp"
#include "../../../nfc-laboratory_amd/csrc/nfc_config.hpp"

extern "C" {

struct sim_frame
{
   uint32_t stream_id, tech_type, frame_type, frame_flags, frame_phase, frame_rate, length, reserved;
   uint64_t sample_start, sample_end, sample_rate;
   uint8_t data[512];
};

/* what the front end and the carrier detector own in the stream state */
struct FrontScalars
{
   uint32_t clock, pulseFilter;
   float env, n1, mdev, avg, edgePeak;
   uint32_t edgeTime, carrierOff, carrierOn;
};

static FrontScalars front_get(const NfcStreamState &s)
{
   return FrontScalars {s.clock, s.pulseFilter, s.env, s.n1, s.mdev, s.avg, s.edgePeak, s.edgeTime, s.carrierOff, s.carrierOn};
}

static void front_put(NfcStreamState &s, const FrontScalars &f)
{
   s.clock = f.clock; s.pulseFilter = f.pulseFilter; s.env = f.env; s.n1 = f.n1; s.mdev = f.mdev; s.avg = f.avg;
   s.edgePeak = f.edgePeak; s.edgeTime = f.edgeTime; s.carrierOff = f.carrierOff; s.carrierOn = f.carrierOn;
}

struct SimStats
{
   uint64_t searchSegments, locks, takeBacks, detectorSteps, samples;
};

long tilemajor_decode(const float *samples, uint64_t count, uint32_t sampleRate, uint32_t lane, uint32_t enabled, uint32_t tile,
                      sim_frame *out, uint32_t cap, SimStats *stats)
{
   NfcHostParams p;
   p.sampleRate = sampleRate;
   p.enabled = enabled;

   NfcConfig cfg;
   if (!nfc_build_config(p, cfg))
      return -1;

   std::vector<float> rings((4 * NFC_HIST + NFC_PROD + cfg.corrTotal) * NFC_LANES, 0.0f);
   std::vector<uint8_t> bytes(NFC_STREAM_BYTES, 0);
   std::vector<uint32_t> arena(1u << 22, 0);

   NfcLaneMem mem;
   mem.linked = false;
   mem.flags = nullptr;
   mem.ring = rings.data();
   mem.lane = lane;
   mem.exact = true; /* ring positions by exact modulo throughout: always right, and independent of the order of the passes */
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

   SimStats st {};
   std::vector<NfcNow> now(tile);
   std::vector<FrontScalars> after(tile);
   std::vector<uint32_t> emitted(tile);

   for (uint64_t base = 0; base < count; base += tile)
   {
      const uint32_t n = (uint32_t)((count - base) < tile ? (count - base) : tile);
      uint32_t pos = 0;

      while (pos < n)
      {
         if (s.lockTech != 0 || s.unlock != 0)
         {
            nfc_step(cfg, s, mem, samples[base + pos], true);
            pos++;
            st.samples++;
            continue;
         }

         /* ---- search segment [pos, n) ---- */
         st.searchSegments++;

         const NfcSearchRegs records0 = s.u.search;
         const uint32_t bank0 = s.bankClock;

         /* the correlation rings of the lane as they are at the start of the segment: a pass that is taken back has
          * written sums for samples the detector never sees (a kernel would hold those writes back instead) */
         std::vector<float> corr0(cfg.corrTotal);
         for (uint32_t i = 0; i < cfg.corrTotal; i++)
            corr0[i] = rings[(size_t)(NFC_R_CORR + i) * NFC_LANES + lane];

         /* (a) front end + carrier detection, speculative */
         for (uint32_t k = pos; k < n; k++)
         {
            ++s.clock;
            ++s.pulseFilter;
            now[k] = nfc_front_end(cfg, s, mem, samples[base + k]);
            nfc_detect_carrier(cfg, s, mem);
            after[k] = front_get(s);
            emitted[k] = ctl[0];
         }

         /* (b) the technologies in bank order */
         uint32_t lockTech = 0, lockAt = n;
         uint32_t ranTo[4] = {pos, pos, pos, pos}; /* first sample a technology has not seen */

         auto run = [&](uint32_t t, uint32_t from, uint32_t to, uint32_t bankBefore) -> bool {
            /* one technology over [from, to); true when it locked (lockAt / lockTech updated) */
            uint32_t bank = bankBefore;
            for (uint32_t k = from; k < to; k++)
            {
               s.clock = after[k].clock;
               s.env = after[k].env;
               ranTo[t] = k + 1;
               const bool armed = s.clock >= 1024u && !(s.env < cfg.powerThreshold);
               if (!armed)
                  continue;
               s.bankClock = bank;
               nfc_advance_positions(cfg, s, mem);
               st.detectorSteps++;
               bool hit = false;
               if (t == 0) { NfcTapsA ta; nfca_load_taps(cfg, s, mem, ta); hit = nfca_detect(cfg, s, mem, ta, now[k]); }
               if (t == 1) { NfcTapsB tb; nfcb_load_taps(cfg, s, mem, tb); hit = nfcb_detect(cfg, s, mem, tb, now[k]); }
               if (t == 2) { NfcTapsF tf; nfcf_load_taps(cfg, s, mem, tf); hit = nfcf_detect(cfg, s, mem, tf, now[k]); }
               if (t == 3) { NfcTapsV tv; nfcv_load_taps(cfg, s, mem, tv); hit = nfcv_detect(cfg, s, mem, tv, now[k]); }
               if (hit)
               {
                  lockAt = k;
                  lockTech = NFC_TECH_A + t;
                  return true;
               }
               bank = s.clock; /* the bank was stepped on this sample and nobody locked (so far as this pass knows) */
            }
            return false;
         };

         for (uint32_t t = 0; t < 4; t++)
         {
            if (!(cfg.enabled & (1u << t)))
               continue;

            /* technologies behind the locking one in bank order do not see the sample of the lock */
            const uint32_t to = lockTech ? lockAt : n;

            if (run(t, pos, to, bank0) && t > 0)
            {
               /* a technology further back in the bank locked before the ones ahead of it stopped: take those back to
                * the lock sample (which they do see) */
               for (uint32_t e = 0; e < t; e++)
               {
                  if (!(cfg.enabled & (1u << e)) || ranTo[e] <= lockAt + 1)
                     continue;
                  st.takeBacks++;
                  if (e == 0) std::memcpy(s.u.search.detA, records0.detA, sizeof(records0.detA));
                  if (e == 1) std::memcpy(s.u.search.detB, records0.detB, sizeof(records0.detB));
                  if (e == 2) std::memcpy(s.u.search.detF, records0.detF, sizeof(records0.detF));
                  {
                     const uint32_t lo = e == 0 ? cfg.corrOffset[0] : (e == 2 ? cfg.corrOffset[3] : 0u);
                     const uint32_t hi = e == 0 ? cfg.corrOffset[3] : (e == 2 ? cfg.corrOffset[5] : 0u);
                     for (uint32_t i = lo; i < hi; i++)
                        rings[(size_t)(NFC_R_CORR + i) * NFC_LANES + lane] = corr0[i];
                  }
                  const uint32_t keepTech = lockTech, keepAt = lockAt;
                  const bool again = run(e, pos, keepAt + 1, bank0);
                  if (again)
                     return -3; /* cannot lock now where it did not before */
                  lockTech = keepTech;
                  lockAt = keepAt;
               }
            }
         }

         /* (c) settle the segment */
         if (lockTech)
         {
            st.locks++;
            front_put(s, after[lockAt]);
            ctl[0] = emitted[lockAt];
            /* bankClock: clock of the last armed sample before the lock */
            uint32_t bank = bank0;
            for (uint32_t k = pos; k < lockAt; k++)
               if (after[k].clock >= 1024u && !(after[k].env < cfg.powerThreshold))
                  bank = after[k].clock;
            s.bankClock = bank;
            nfc_advance_positions(cfg, s, mem);
            nfc_enter_lock(s, mem, lockTech);
            st.samples += lockAt + 1 - pos;
            pos = lockAt + 1;
         }
         else
         {
            front_put(s, after[n - 1]);
            uint32_t bank = bank0;
            for (uint32_t k = pos; k < n; k++)
               if (after[k].clock >= 1024u && !(after[k].env < cfg.powerThreshold))
                  bank = after[k].clock;
            s.bankClock = bank;
            nfc_advance_positions(cfg, s, mem);
            st.samples += n - pos;
            pos = n;
         }
      }
   }

   if (stats)
      *stats = st;

   if (ctl[1])
      return -2;

   long frames = 0;
   uint32_t at = 0;
   while (at < ctl[0])
   {
      const uint32_t *w = arena.data() + at + 1;
      uint32_t len = w[7];
      if (frames < cap)
      {
         sim_frame &f = out[frames];
         std::memset(&f, 0, sizeof(f));
         f.tech_type = w[0]; f.frame_type = w[1]; f.frame_flags = w[2]; f.frame_phase = w[3];
         f.frame_rate = w[4]; f.sample_start = w[5]; f.sample_end = w[6]; f.length = len;
         f.sample_rate = sampleRate;
         std::memcpy(f.data, w + 8, len);
      }
      frames++;
      at += NFC_FRAME_HEADER_WORDS + ((len + 3) >> 2);
   }
   return frames;
}

}
