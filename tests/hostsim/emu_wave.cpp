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

/* the history is as deep as the stored one; the samples a tile written ahead displaces are kept beside it (nfc_wave.hpp) */
#define NFC_X_OLD_INDEX(mem, sampleClock) nfc_wave_x_old_index((mem).ring, (sampleClock))
#define NFC_RING_STRIDE 1u
#define NFC_WAVE_LDS
#define NFC_RING_FLOAT float
#define NFC_HIST_F 256u
struct NfcWaveDeep;
#define NFC_F_DEEP_CTX const NfcWaveDeep *
#define NFC_F_DEEP(mem, region, clk) nfc_wave_f_deep((mem), (region), (clk))
template <class Mem>
static inline float nfc_wave_f_deep(const Mem &mem, uint32_t region, uint32_t clk);

#define NFC_DEV static inline
static inline uint32_t emu_add(uint32_t *p, uint32_t v) { uint32_t old = *p; *p += v; return old; }
#define NFC_ATOMIC_ADD(ptr, value) emu_add((ptr), (value))
#define NFC_ANY(predicate) (predicate)
#include "../../nfc-laboratory_amd/csrc/nfc_types.h"

/* index, from the start of the wave's ring storage, of the raw sample of clock `clk`: in the history, or among the samples
 * the tile written ahead has displaced (layout: nfc_wave.hpp, NFC_WAVE_XOLD) */
static inline uint32_t nfc_wave_x_old_index(const NFC_RING_FLOAT *ring, uint32_t clk)
{
   const uint32_t displaced = NFC_HIST + 3u * NFC_HIST_F + NFC_PROD + NFC_CORR_MAX;
   const uint32_t clock0 = __builtin_bit_cast(uint32_t, (float)ring[displaced + NFC_LANES]);
   const uint32_t k = clk - (clock0 - (NFC_HIST - 1u)); /* sample clock0 - 511 + j was displaced by tile sample j */
   return k < NFC_LANES ? displaced + k : (clk & (NFC_HIST - 1u));
}

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
/* (diagnostic: at the samples at which a piece may hand over - its successors' verify samples - why it does not; printed with
 * NFC_EMU_WAVE_STATS) */
extern uint64_t emu_handover_counts[8][4];
static inline void emu_handover_tally(const NfcStreamState &s, const NfcStreamCold &cold, uint32_t ran);
#define NFC_HANDOVER_TALLY(s, cold, ran) emu_handover_tally((s), (cold), (ran))
#include "../../nfc-laboratory_amd/csrc/nfc_scan.hpp"
static inline void emu_handover_tally(const NfcStreamState &s, const NfcStreamCold &cold, uint32_t ran)
{
   const int why = s.lockTech ? 1 + (int)(((s.lockTech & 0xFFu) - 1u) & 3u)
                              : (s.unlock ? 5 : (s.bankClock != s.clock ? 6 : ((uint32_t)(s.clock - cold.bankRun) < NFC_WINDOW_STEADY ? 7 : 0)));
   emu_handover_counts[why][ran < 40000u ? 0 : (ran < 100000u ? 1 : (ran < 250000u ? 2 : 3))]++;
}

#define NFC_WAVE_LANE() (wavesim::lane())
#define NFC_WAVE_BARRIER() wavesim::barrier()
#define NFC_WAVE_BALLOT(p) wavesim::ballot(p)
#define NFC_WAVE_UNIFORM_BEGIN if (wavesim::lane() == 0) {
#define NFC_WAVE_UNIFORM_END } wavesim::barrier();
#define NFC_WAVE_UNIFORM_U32(x) ((uint32_t)(x))
#define NFC_WAVE_UNIFORM_LEAVE(slot, value) ((slot) = (value))
#define NFC_WAVE_UNIFORM_TAKE(slot, value) ((value) = (slot))
#define NFC_WAVE_UNIFORM_TAKE_BITS(slot, value) ((value) = __builtin_bit_cast(uint32_t, (float)(slot)))
#define NFC_WAVE_READ_FENCE() wavesim::barrier()
#define NFC_WAVE_PICK_F(reg, array, j) ((array)[(j)])
#define NFC_WAVE_PICK_U32(reg, array, j) (wavesim::shfl((reg), (j)))
#define NFC_WAVE_SHFL_F(reg, j) (wavesim::shfl((reg), (j)))
#define NFC_WAVE_CONFIG(cfgPtr, lds, cc) ((cc) = *(cfgPtr))
#define NFC_WAVE_NOINLINE static __attribute__((noinline))
#define NFC_WAVE_STAT_ADD(p, v) (*(p) += (v))
#define NFC_WAVE_STAT_MAX(p, v) (*(p) = *(p) > (v) ? *(p) : (v))

static inline float emu_scan_add(float v)
{
   const uint32_t lane = wavesim::lane();
   for (uint32_t d = 1; d < 64; d <<= 1)
   {
      const float t = wavesim::shfl(v, lane >= d ? lane - d : lane);
      if (lane >= d)
         v += t;
   }
   return v;
}

static inline float emu_max(float v)
{
   for (uint32_t d = 1; d < 64; d <<= 1)
   {
      const float t = wavesim::shfl(v, wavesim::lane() ^ d);
      v = t > v ? t : v;
   }
   return v;
}

#define NFC_WAVE_SCAN_ADD_F(v) emu_scan_add(v)
#define NFC_WAVE_MAX_F(v) emu_max(v)

/* NFC_EMU_WAVE_VERIFY=1: every tile is decoded twice - with the bulk paths and sample by sample - and everything the two
 * leave behind (decoder state, rings, protocol state, frame bytes) is compared bit for bit */
#include "../../nfc-laboratory_amd/csrc/nfc_scan_launch.h"
struct NfcWaveLds;
struct NfcWaveItem;
struct NfcWaveSink;
struct NfcWaveFetch;
static void emu_verify_tile(const NfcConfig *cfgPtr, const NfcConfig &cc, const NfcScanArgs &A, const NfcWaveItem &it, NfcWaveLds *lds, const NfcWaveSink &sink,
                            uint32_t n, uint32_t pos, bool carry, uint32_t warmFront, uint32_t warm, const NfcWaveFetch &fetched);
#define NFC_WAVE_TILE_HOOK emu_verify_tile

extern uint64_t emu_wave_counts[64][2];
static bool emu_counting = true;
static uint32_t emu_trace_clock;
#define NFC_WAVE_COUNT_DETECTORS(b) do { emu_wave_counts[32 + (b)][0]++; if (std::getenv("NFC_EMU_TRACE_SEARCH")) std::fprintf(stderr, "[trace] %u %u\n", clock0 + 1u + from, (unsigned)(b)); } while (0)
#define NFC_WAVE_COUNT_NOT_TAKEN(key, grid) do { if (wavesim::lane() == 0 && std::getenv("NFC_EMU_TRACE_NOT_TAKEN")) std::fprintf(stderr, "[not taken] key %u grid %d clock %u\n", (unsigned)(key), (int)(grid), clock0 + 1u + from); } while (0)
/* (diagnostic, -DNFC_WAVE_COUNT_VISITS: what brings the wave to the NFC-B 106k detector's record) */
extern uint64_t emu_b_counts[16];
#define NFC_WAVE_COUNT_B(c, m, I, clk, env, edge, deep) do { if (wavesim::lane() == 0 && emu_counting) { \
   const bool reset_ = (deep) > (c).maxDepth[1] || ((m).auxTime && (clk) > (m).auxTime + (c).b[I].p1); \
   const int stage_ = !(m).symStart ? 0 : (!(m).symEnd ? 1 : 2); \
   int why_ = reset_ ? 0 : ((clk) == (m).winEnd ? 1 : (stage_ && (clk) < (m).winStart ? 2 : 3)); \
   emu_b_counts[stage_ * 4 + why_]++; } } while (0)
/* (diagnostic, -DNFC_WAVE_COUNT_VISITS: visits of one detector alone at consecutive samples) */
static uint32_t emu_run_clock, emu_run_here;
extern uint64_t emu_run_counts[9][2];
uint64_t emu_b_counts[16];
uint64_t emu_handover_counts[8][4];
#define NFC_WAVE_COUNT_RUN(clk, here) do { if (wavesim::lane() == 0 && emu_counting) { const uint32_t h_ = (here); const bool single_ = h_ && !(h_ & (h_ - 1u)); \
   if (single_) { const int b_ = __builtin_ctz(h_); emu_run_counts[b_][0]++; if (emu_run_here == h_ && emu_run_clock + 1u == (clk)) emu_run_counts[b_][1]++; } \
   emu_run_clock = (clk); emu_run_here = h_; } } while (0)
#define NFC_WAVE_COUNT(key, which, count) do { if (wavesim::lane() == 0 && emu_counting) emu_wave_counts[(key) & 63u][(which)] += (count); } while (0)

#define NFC_WAVE_DEBUG_FETCH(f, clock) do { if (std::getenv("NFC_EMU_DEBUG_FETCH") && (wavesim::lane() < 2) && (clock) >= 131071u && (clock) < 131300u) std::fprintf(stderr, "[fetch] lane %u clock %u &f %p env %g x %g\n", wavesim::lane(), (unsigned)(clock), (const void *)&(f), (f).env, (f).x); } while (0)
/* NFC_EMU_WATCH=<index>: report the phase of the tile loop after which ring[index] first differs from what it held when the tile began */
static int emu_watch_index = -2;
static float emu_watch_value;
static bool emu_watch_armed;
#define NFC_WAVE_DEBUG_POINT(lds, tag) do { if (emu_watch_index == -2) { const char *w = std::getenv("NFC_EMU_WATCH"); emu_watch_index = w ? std::atoi(w) : -1; } \
   if (emu_watch_index >= 0 && emu_watch_armed && std::memcmp(&(lds)->ring[emu_watch_index], &emu_watch_value, 4) != 0) { \
      std::fprintf(stderr, "[watch] ring[%d] changed to %g (was %g) after phase %u (1 bulk call, 2 search step, 3 step, 4 load, 5 end), fibre %u at %u clock %u key %u\n", emu_watch_index, \
                   (lds)->ring[emu_watch_index], emu_watch_value, (unsigned)(tag), wavesim::lane(), (lds)->u.at, (lds)->u.s.clock, (lds)->u.key); emu_watch_value = (lds)->ring[emu_watch_index]; } } while (0)
#include <cstring>
#define NFC_WAVE_DEBUG_FSTART(t, d, m, sd) do { static const bool on = std::getenv("NFC_EMU_DEBUG_FSTART") != nullptr; static unsigned long shown = 0; \
   const bool hit = (t) >= (d).guardEnd && ((t) == (d).guardEnd || (t) > (d).waitingEnd || ((t) >= (m).winStart && (((sd) >= (m).thr && (sd) > (m).peak) || (t) == (m).sync || (t) == (m).winEnd))); \
   if (on && hit && wavesim::lane() == lds->u.at && (shown++ % 997) < 3 && shown < 200000) std::fprintf(stderr, "[fstart] t %u guardEnd %u waitingEnd %u winStart %u winEnd %u sync %u sd %g thr %g peak %g stage %u pulses %u symStart %u symEnd %u\n", \
   (t), (d).guardEnd, (d).waitingEnd, (m).winStart, (m).winEnd, (m).sync, (double)(sd), (double)(m).thr, (double)(m).peak, (m).stage, (m).pulses, (m).symStart, (m).symEnd); } while (0)
#include "../../nfc-laboratory_amd/csrc/nfc_wave.hpp"

static int emu_verify_mode()
{
   static int mode = -1;
   if (mode < 0)
   {
      const char *v = std::getenv("NFC_EMU_WAVE_VERIFY");
      mode = v && v[0] ? std::atoi(v) : 0;
   }
   return mode;
}

uint64_t emu_run_counts[9][2];
uint64_t emu_wave_counts[64][2]; /* per stage key: samples committed in bulk, samples stepped */

namespace {
struct CountPrinter
{
   ~CountPrinter()
   {
      if (!std::getenv("NFC_EMU_WAVE_STATS"))
         return;
      static const char *names[] = {"none", "search", "upkeep", "unarmed", "A poll", "A ask start", "A ask symbol", "A bpsk start", "A bpsk symbol", "B poll",
                                    "B start", "B symbol", "F data", "F start", "V poll", "V start", "V symbol"};
      static const char *det[] = {"A106", "A212", "A424", "B106", "B212", "F212", "F424", "V"};
      for (uint32_t k = 0; k < 8; k++)
         if (emu_wave_counts[32 + k][0])
            std::fprintf(stderr, "[emu wave] search stepped for %-5s %10llu\n", det[k], (unsigned long long)emu_wave_counts[32 + k][0]);
      std::fprintf(stderr, "[emu wave] NFC-B detectors stepped on their own %llu, steps in the wake of another %llu, bulk paths not taken %llu, unarmed / carrier steps %llu\n",
                   (unsigned long long)emu_wave_counts[44][0], (unsigned long long)emu_wave_counts[45][0], (unsigned long long)emu_wave_counts[46][0],
                   (unsigned long long)emu_wave_counts[47][0]);
      {
         static const char *why[] = {"hands over", "locked NFC-A", "locked NFC-B", "locked NFC-F", "locked NFC-V", "unlock pending", "bank not stepped", "steady run too short"};
         for (uint32_t k = 0; k < 8; k++)
            if (emu_handover_counts[k][0] | emu_handover_counts[k][1] | emu_handover_counts[k][2] | emu_handover_counts[k][3])
               std::fprintf(stderr, "[emu wave] at a successor's verify sample, %-22s pieces that had run < 40 k: %6llu, < 100 k: %6llu, < 250 k: %6llu, longer: %6llu\n", why[k],
                            (unsigned long long)emu_handover_counts[k][0], (unsigned long long)emu_handover_counts[k][1], (unsigned long long)emu_handover_counts[k][2], (unsigned long long)emu_handover_counts[k][3]);
      }
      for (uint32_t k = 0; k < 12; k++)
         if (emu_b_counts[k])
            std::fprintf(stderr, "[emu wave] B106 shown its record in stage %u because of %s: %llu\n", k / 4, (k % 4) == 0 ? "a reset" : ((k % 4) == 1 ? "the window's end" : ((k % 4) == 2 ? "an edge before the window" : "a new extreme")), (unsigned long long)emu_b_counts[k]);
      for (uint32_t k = 0; k < 8; k++)
         if (emu_run_counts[k][0])
            std::fprintf(stderr, "[emu wave] visits of %-5s alone %10llu, of them right after a visit of the same alone %10llu\n", det[k], (unsigned long long)emu_run_counts[k][0], (unsigned long long)emu_run_counts[k][1]);
      for (uint32_t k = 0; k < 9; k++)
         if (emu_wave_counts[51 + k][0])
            std::fprintf(stderr, "[emu wave] shown in place: %-5s %10llu\n", k < 8 ? det[k] : "B only", (unsigned long long)emu_wave_counts[51 + k][0]);
      std::fprintf(stderr, "[emu wave] NFC-F listen tracker applied in place %llu, NFC-F detectors stepped on their own %llu\n", (unsigned long long)emu_wave_counts[49][0],
                   (unsigned long long)emu_wave_counts[50][0]);
      std::fprintf(stderr, "[emu wave] bulk-path calls %llu, search values formed %llu, locked values formed %llu (with walked sums: %llu), tiles %llu\n", (unsigned long long)emu_wave_counts[40][0],
                   (unsigned long long)emu_wave_counts[41][0], (unsigned long long)emu_wave_counts[42][0], (unsigned long long)emu_wave_counts[48][0], (unsigned long long)emu_wave_counts[43][0]);
      for (uint32_t k = 0; k < 17; k++)
         if (emu_wave_counts[k][0] | emu_wave_counts[k][1])
            std::fprintf(stderr, "[emu wave] %-14s bulk %12llu stepped %10llu\n", names[k], (unsigned long long)emu_wave_counts[k][0],
                         (unsigned long long)emu_wave_counts[k][1]);
   }
} countPrinter;
}

static void emu_verify_tile(const NfcConfig *cfgPtr, const NfcConfig &cc, const NfcScanArgs &A, const NfcWaveItem &it, NfcWaveLds *lds, const NfcWaveSink &sink,
                            uint32_t n, uint32_t pos, bool carry, uint32_t warmFront, uint32_t warm, const NfcWaveFetch &fetched)
{
   const int mode = emu_verify_mode();

   if (mode == 2)
   {
      nfc_wave_tile(cfgPtr, cc, A, it, lds, sink, n, pos, carry, warmFront, warm, fetched, false); /* no bulk paths at all */
      return;
   }

   if (mode != 1)
   {
      nfc_wave_tile(cfgPtr, cc, A, it, lds, sink, n, pos, carry, warmFront, warm, fetched, true);
      return;
   }

   /* shared by the 64 fibres of the wave (they run one after the other) */
   static NfcWaveLds savedLds, fastLds;
   static uint32_t dummySink[4096], dummyCtl[2];
   static uint32_t keyBefore;

   wavesim::barrier();
   if (wavesim::lane() == 0)
   {
      savedLds = *lds;
      keyBefore = nfc_wave_stage(lds->u.s, lds->u.consumed < warm);
   }
   wavesim::barrier();

   if (wavesim::lane() == 0 && emu_watch_index >= 0 && !std::getenv("NFC_EMU_WATCH_STEPPED"))
   {
      emu_watch_value = lds->ring[emu_watch_index];
      emu_watch_armed = true;
   }
   nfc_wave_tile(cfgPtr, cc, A, it, lds, sink, n, pos, carry, warmFront, warm, fetched, true);
   wavesim::barrier();
   NFC_WAVE_DEBUG_POINT(lds, 5u);
   wavesim::barrier();
   emu_watch_armed = false;

   wavesim::barrier();
   if (wavesim::lane() == 0)
   {
      fastLds = *lds;
      *lds = savedLds;
      dummyCtl[0] = 0;
      dummyCtl[1] = 0;
      /* (the lane's frame records are chained through the sink: a record emitted into the dummy sink must not be linked
       * to one whose place is a place in the real sink - nfc_emit would write the link far outside the dummy) */
      lds->cold.frameHead = 0;
      lds->cold.frameTail = 0;
   }
   wavesim::barrier();

   /* again, sample by sample, frames into a dummy sink */
   NfcWaveSink quiet = sink;
   quiet.words = dummySink;
   quiet.ctl = dummyCtl;
   quiet.capacity = 4096;

   emu_counting = false;
   if (wavesim::lane() == 0 && emu_watch_index >= 0 && std::getenv("NFC_EMU_WATCH_STEPPED"))
   {
      emu_watch_value = lds->ring[emu_watch_index];
      emu_watch_armed = true;
   }
   nfc_wave_tile(cfgPtr, cc, A, it, lds, quiet, n, pos, carry, warmFront, warm, fetched, false);
   emu_watch_armed = false;
   wavesim::barrier();
   emu_counting = true;

   wavesim::barrier();

   if (wavesim::lane() == 0)
   {
      /* the frame records are chained by their place in the sink */
      NfcStreamCold ca = fastLds.cold, cb = lds->cold;
      ca.frameHead = cb.frameHead = 0;
      ca.frameTail = cb.frameTail = 0;

      const NfcStreamState &a = fastLds.u.s, &b = lds->u.s;

      const bool same = std::memcmp(&a, &b, sizeof(a)) == 0 && fastLds.u.at == lds->u.at && std::memcmp(&ca, &cb, sizeof(ca)) == 0 &&
                        std::memcmp(fastLds.ring, lds->ring, sizeof(float) * NFC_R_PROD) == 0 &&
                        std::memcmp(fastLds.ring + NFC_R_CORR, lds->ring + NFC_R_CORR, sizeof(float) * NFC_CORR_MAX) == 0 &&
                        std::memcmp(fastLds.bytes, lds->bytes, NFC_STREAM_BYTES) == 0 && fastLds.flags == lds->flags;

      if (!same)
      {
         std::fprintf(stderr, "[emu wave verify] tile at stream position %u (clock %u..), lane slot %u, stage key before %u: bulk and stepped results differ\n", pos,
                      savedLds.u.s.clock + 1u, it.w, keyBefore);
         const uint32_t *pa = (const uint32_t *)&a, *pb = (const uint32_t *)&b;
         for (uint32_t i = 0; i < sizeof(NfcStreamState) / 4; i++)
            if (pa[i] != pb[i])
               std::fprintf(stderr, "   state word %u: bulk %08x stepped %08x\n", i, pa[i], pb[i]);
         const uint32_t *qa = (const uint32_t *)&ca, *qb = (const uint32_t *)&cb;
         for (uint32_t i = 0; i < sizeof(NfcStreamCold) / 4; i++)
            if (qa[i] != qb[i])
               std::fprintf(stderr, "   cold word %u: bulk %08x stepped %08x\n", i, qa[i], qb[i]);
         for (uint32_t i = 0; i < NFC_CORR_MAX; i++)
            if (std::memcmp(&fastLds.ring[NFC_R_CORR + i], &lds->ring[NFC_R_CORR + i], 4) != 0)
               std::fprintf(stderr, "   corr ring %u: bulk %g stepped %g\n", i, fastLds.ring[NFC_R_CORR + i], lds->ring[NFC_R_CORR + i]);
         std::fprintf(stderr, "   at: bulk %u stepped %u; n %u; env[0] %g %g env[n-1] %g %g; fetched.env (fibre 0) %g; stepped count %u %u\n", fastLds.u.at, lds->u.at, n, fastLds.env[0], lds->env[0],
                      fastLds.env[n - 1], lds->env[n - 1], fetched.env, fastLds.u.stepped, lds->u.stepped);
         for (uint32_t i = 0; i < NFC_R_PROD; i++)
            if (std::memcmp(&fastLds.ring[i], &lds->ring[i], 4) != 0)
               std::fprintf(stderr, "   history ring %u (region %u slot %u): bulk %g stepped %g\n", i, i / NFC_HIST, i % NFC_HIST, fastLds.ring[i], lds->ring[i]);
         for (uint32_t i = 0; i < NFC_STREAM_BYTES; i++)
            if (fastLds.bytes[i] != lds->bytes[i])
               std::fprintf(stderr, "   frame byte %u: bulk %02x stepped %02x\n", i, fastLds.bytes[i], lds->bytes[i]);
         for (uint32_t i = 0; i < n; i++)
            if (fastLds.env[i] != lds->env[i])
               std::fprintf(stderr, "   env[%u]: bulk %g stepped %g\n", i, fastLds.env[i], lds->env[i]);
         if (fastLds.flags != lds->flags)
            std::fprintf(stderr, "   flags: bulk %08x stepped %08x\n", fastLds.flags, lds->flags);
         std::abort();
      }

      /* go on from the first run (its frames are the ones in the sink) */
      const uint32_t stepped = fastLds.u.stepped;
      *lds = fastLds;
      lds->u.stepped = stepped;
   }

   wavesim::barrier();
}

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
