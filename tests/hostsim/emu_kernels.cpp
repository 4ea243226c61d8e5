/*
 * TEST INFRASTRUCTURE ONLY - CPU twins of the kernels of nfc_kernels.hip for the emulated build of the host runtime
 * (tests/hostsim/build_emulated.sh). They keep what the host runtime relies on: which stream slots a launch covers, which
 * rows of the work table / uniform layout they read, which of the common / exact-modulo kernels takes a stream block
 * (decided from the device-side clocks exactly like nfc_demod_body), where state, rings and frames go. The per-sample work
 * is the product's device step machine (nfc_core.hpp) run lane after lane instead of 64 lanes in lock step.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <vector>

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
#include "../../nfc-laboratory_amd/csrc/nfc_scan.h"
static void emu_chain_trace(uint32_t lane, const NfcCarry &assumed, const NfcCarry &have, const NfcCarry &left)
{
   if (!std::getenv("NFC_EMU_DEBUG3"))
      return;
   std::fprintf(stderr, "[emu] lane %u assumed!=have:", lane);
#define F(f) if (assumed.f != have.f) std::fprintf(stderr, " " #f " %u->%u (left %u)", (unsigned)assumed.f, (unsigned)have.f, (unsigned)left.f)
   F(chainedA); F(carrierOn); F(carrierOff); F(emitClock); F(emitValid); F(pulsesF[0]); F(pulsesF[1]);
   for (int t = 0; t < 4; t++) { F(tim[t].lastCommand); F(tim[t].maxFrameSize); F(tim[t].protoGuardTime); F(tim[t].protoWaitingTime); }
#undef F
   if (assumed.thrF[0] != have.thrF[0]) std::fprintf(stderr, " thrF0 %g->%g (left %g)", assumed.thrF[0], have.thrF[0], left.thrF[0]);
   if (assumed.thrF[1] != have.thrF[1]) std::fprintf(stderr, " thrF1 %g->%g (left %g)", assumed.thrF[1], have.thrF[1], left.thrF[1]);
   if (assumed.edgeTime != have.edgeTime) std::fprintf(stderr, " edgeTime %u->%u", assumed.edgeTime, have.edgeTime);
   std::fprintf(stderr, " [emit %u/%u vs %u/%u]", assumed.emitValid, assumed.emitClock, have.emitValid, have.emitClock);
   {
      const uint32_t *p = (const uint32_t *)&assumed.search, *q = (const uint32_t *)&have.search;
      for (uint32_t i = 0; i < sizeof(NfcSearchRegs) / 4; i++)
         if (p[i] != q[i]) std::fprintf(stderr, " search[%u] %x->%x", i, p[i], q[i]);
   }
   std::fprintf(stderr, "\n");
}
#define NFC_CHAIN_TRACE(lane, a, b, c) emu_chain_trace((lane), (a), (b), (c))
/* NFC_EMU_CHAIN_STATS=1: how often the NFC-F pulse memory a lane meets is what it assumed / is clear */
static unsigned long emuChainStats[8];
static void emu_chain_stats(const NfcCarry &assumed, const NfcCarry &have, uint32_t used)
{
   static const bool on = std::getenv("NFC_EMU_CHAIN_STATS") != nullptr;
   if (!on)
      return;
   for (int i = 0; i < 2; i++)
   {
      const bool same = assumed.pulsesF[i] == have.pulsesF[i] && std::memcmp(&assumed.thrF[i], &have.thrF[i], 4) == 0;
      const bool clear = have.pulsesF[i] == 0 && have.thrF[i] == 0.0f;
      emuChainStats[0]++;
      emuChainStats[1] += same;
      emuChainStats[2] += clear;
      emuChainStats[3] += ((used >> (12 + i)) & 1u) ? 1 : 0;
      emuChainStats[4] += (((used >> (12 + i)) & 1u) && same) ? 1 : 0;
      emuChainStats[5] += (((used >> (12 + i)) & 1u) && clear) ? 1 : 0;
   }
   if ((emuChainStats[0] % 2000) == 0)
      std::fprintf(stderr, "[emu chain stats] lane x detector meetings %lu: memory as assumed %lu, clear %lu; of %lu that looked at it: as assumed %lu, clear %lu\n", emuChainStats[0],
                   emuChainStats[1], emuChainStats[2], emuChainStats[3], emuChainStats[4], emuChainStats[5]);
}
#define NFC_CHAIN_STATS(a, b, u) emu_chain_stats((a), (b), (u))
static void emu_seam_debug(uint32_t k, const NfcScanPoint &start, const NfcScanPoint &end, uint32_t edge)
{
   if (!std::getenv("NFC_EMU_SEAM_DEBUG"))
      return;
   std::fprintf(stderr, "[emu seam] chunk %u:%s%s%s%s%s%s%s%s | env %.7g vs %.7g  pf %u vs %u\n", k, start.env != end.env ? " env" : "", start.n1 != end.n1 ? " n1" : "",
                start.mdev != end.mdev ? " mdev" : "", start.avg != end.avg ? " avg" : "", start.edgePeak != end.edgePeak ? " edgePeak" : "",
                start.pulseFilter != end.pulseFilter ? " pulseFilter" : "", ((start.zone ^ end.zone) & 0xFFu) ? " zone" : "",
                ((start.zone & 0x100u) && start.edgeTime != edge) ? " edgeTime" : "", start.env, end.env, start.pulseFilter, end.pulseFilter);
}
#define NFC_SEAM_DEBUG(k, a, b, e) emu_seam_debug((k), (a), (b), (e))
static void emu_carry_debug(const NfcCarry &a, const NfcCarry &b, bool meeting, uint32_t tracked, uint32_t used);
#define NFC_CARRY_DEBUG(a, b, m, t, u) emu_carry_debug((a), (b), (m), (t), (u))
#include "../../nfc-laboratory_amd/csrc/nfc_scan.hpp"
#include "../../nfc-laboratory_amd/csrc/nfc_launch.h"
#include "../../nfc-laboratory_amd/csrc/nfc_scan_launch.h"
#include "../../nfc-laboratory_amd/csrc/nfc_envelope.hpp"

namespace fakehip {
dim3 launchGrid, launchBlock;
std::recursive_mutex launchMutex;
thread_local int currentDevice = -1;
}

/* NFC_EMU_CARRY_TALLY=1: which part of an assumption was wrong where nfc_carry_same fails (printed when the library is unloaded) */
static unsigned long emuCarryTally[16];
static struct EmuCarryTallyPrinter
{
   ~EmuCarryTallyPrinter()
   {
      if (emuCarryTally[0])
         std::fprintf(stderr, "[emu carry tally] failures %lu: chainedA %lu, carrier on/off %lu, edge time %lu, last command %lu, frame size %lu, guard time %lu, waiting time %lu, "
                              "NFC-F pulse memory %lu, records %lu (A %lu, B %lu, F %lu, V %lu); one cause only: %lu\n", emuCarryTally[0], emuCarryTally[1], emuCarryTally[2], emuCarryTally[3],
                      emuCarryTally[4], emuCarryTally[5], emuCarryTally[6], emuCarryTally[7], emuCarryTally[8], emuCarryTally[9], emuCarryTally[10], emuCarryTally[11], emuCarryTally[12],
                      emuCarryTally[13], emuCarryTally[14]);
   }
} emuCarryTallyPrinter;

static void emu_carry_debug(const NfcCarry &a, const NfcCarry &b, bool meeting, uint32_t tracked, uint32_t used)
{
   static const bool tally = std::getenv("NFC_EMU_CARRY_TALLY") != nullptr;
   if (tally)
   {
      const uint32_t matters = used & ~(used >> 22) & 0xFu;
      unsigned causes = 0;
      auto count = [&](int k, bool wrong) { if (wrong) { emuCarryTally[k]++; causes++; } };
      emuCarryTally[0]++;
      count(1, (matters & 1u) && a.chainedA != b.chainedA);
      count(2, (a.carrierOn != 0) != (b.carrierOn != 0) || (a.carrierOff != 0) != (b.carrierOff != 0));
      count(3, !(meeting ? a.edgeTime == b.edgeTime : nfc_edge_time(a, tracked) == nfc_edge_time(b, tracked)));
      bool cmd = false, size = false, guard = false, wait = false;
      for (int t = 0; t < 4; t++)
         if ((matters >> t) & 1u)
         {
            cmd = cmd || (((used >> (4 + t)) & 1u) && a.tim[t].lastCommand != b.tim[t].lastCommand);
            size = size || a.tim[t].maxFrameSize != b.tim[t].maxFrameSize;
            guard = guard || a.tim[t].protoGuardTime != b.tim[t].protoGuardTime;
            wait = wait || a.tim[t].protoWaitingTime != b.tim[t].protoWaitingTime;
         }
      count(4, cmd); count(5, size); count(6, guard); count(7, wait);
      bool pulses = false;
      for (int i = 0; i < 2; i++)
         pulses = pulses || (((used >> (12 + i)) & 1u) && (a.pulsesF[i] != b.pulsesF[i] || std::memcmp(&a.thrF[i], &b.thrF[i], 4) != 0));
      count(8, pulses);
      count(9, !nfc_records_same(a.search, b.search, used));
      {
         NfcSearchRegs x = a.search, y = b.search;
         nfc_records_canonical(x);
         nfc_records_canonical(y);
         emuCarryTally[10] += std::memcmp(x.detA, y.detA, sizeof(x.detA)) != 0;
         emuCarryTally[11] += std::memcmp(x.detB, y.detB, sizeof(x.detB)) != 0;
         bool f = false;
         for (int i = 0; i < 2; i++)
            f = f || (((used >> (14 + i)) & 1u) && std::memcmp(&x.detF[i], &y.detF[i], sizeof(x.detF[i])) != 0);
         emuCarryTally[12] += f;
         emuCarryTally[13] += std::memcmp(&x.detV, &y.detV, sizeof(x.detV)) != 0;
      }
      emuCarryTally[14] += causes == 1;
   }
   if (!std::getenv("NFC_EMU_DEBUG3"))
      return;
   std::fprintf(stderr, "[emu] carry_same fails: chained %d on %d off %d edge %d (meeting %d tracked %u: %u vs %u | %u vs %u) records %d\n",
                a.chainedA == b.chainedA, (a.carrierOn != 0) == (b.carrierOn != 0), (a.carrierOff != 0) == (b.carrierOff != 0),
                meeting ? a.edgeTime == b.edgeTime : nfc_edge_time(a, tracked) == nfc_edge_time(b, tracked), (int)meeting, tracked, a.edgeTime, b.edgeTime,
                nfc_edge_time(a, tracked), nfc_edge_time(b, tracked), (int)nfc_records_same(a.search, b.search));
}

namespace {

/* nfc_kernels.hip: nfc_exact_span */
bool exact_span(uint32_t clock, uint32_t count)
{
   const uint32_t start = clock + 1u + 1024u;
   const uint32_t untilWrap = 0u - start;
   return count != 0 && (start < 2048u || untilWrap < count);
}

struct Row
{
   const uint8_t *data;
   uint32_t count;
};

/* nfc_kernels.hip: nfc_row + the lane's own count */
Row row_of(const NfcLaunch &L, uint32_t slot)
{
   Row r {nullptr, 0};
   if (slot >= L.firstSlot && slot < L.firstSlot + L.slotCount)
   {
      if (L.works)
      {
         r.data = L.works[slot].data;
         r.count = L.works[slot].count;
      }
      else
      {
         r.data = L.uniformBase + (uint64_t)(slot - L.firstSlot) * L.uniformPitch;
         r.count = L.uniformCount;
      }
   }
   return r;
}

void demod(const NfcConfig *cfgPtr, const NfcLaunch &L, bool exactKernel)
{
   for (uint32_t b = 0; b < fakehip::launchGrid.x; b++)
   {
      const uint32_t block = L.firstBlock + b;
      uint32_t longest = 0;
      bool anyExact = false;
      bool served = false;

      for (uint32_t lane = 0; lane < NFC_LANES; lane++)
      {
         const uint32_t slot = block * NFC_LANES + lane;
         const Row r = row_of(L, slot);
         longest = r.count > longest ? r.count : longest;
         anyExact = anyExact || exact_span(L.states[slot].clock, r.count);
         served = served || (r.count != 0 && L.states[slot].served == L.launchSeq);
      }

      if (longest == 0 || served)
         continue;

      if ((L.forceExact != 0 || anyExact) != exactKernel)
         continue;

      for (uint32_t lane = 0; lane < NFC_LANES; lane++)
      {
         const uint32_t slot = block * NFC_LANES + lane;
         const Row r = row_of(L, slot);

         if (!r.count)
            continue;

         NfcStreamState s = L.states[slot];

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

         const float *p = reinterpret_cast<const float *>(r.data);

         for (uint32_t k = 0; k < r.count; k++)
         {
            float v;
            if (L.uniformStride == 2)
            {
               volatile float ii = p[2 * k] * p[2 * k];
               volatile float qq = p[2 * k + 1] * p[2 * k + 1];
               v = __builtin_sqrtf(ii + qq);
            }
            else
               v = p[k];

            nfc_step(*cfgPtr, s, mem, v, exactKernel);
         }

         s.served = L.launchSeq;
         L.states[slot] = s;
      }
   }
}

}

void nfc_demod_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) { demod(cfgPtr, L, false); }
void nfc_demod_fixed_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) { demod(cfgPtr, L, false); }
void nfc_demod_exact_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) { demod(cfgPtr, L, true); }
void nfc_demod_fixed_exact_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L) { demod(cfgPtr, L, true); }

void nfc_init_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, uint32_t keepFrontEnd)
{
   for (uint32_t idx = 0; idx < L.slotCount; idx++)
   {
      const uint32_t slot = L.firstSlot + idx;
      const uint32_t block = slot / NFC_LANES;
      const uint32_t lane = slot % NFC_LANES;

      NfcStreamState s = L.states[slot];
      NfcStreamCold cold;
      nfc_state_init(*cfgPtr, s, cold, keepFrontEnd != 0);
      L.states[slot] = s;
      L.cold[slot] = cold;

      float *ring = L.rings + (uint64_t)block * L.ringBlockFloats + lane;
      const uint32_t from = keepFrontEnd ? 4 * NFC_HIST : 0;
      const uint32_t total = L.ringBlockFloats / NFC_LANES;

      for (uint32_t i = from; i < total; i++)
         ring[i * NFC_LANES] = 0.0f;
   }
}

void nfc_magnitude_kernel(const float2 *__restrict__ iq, float *__restrict__ out, uint64_t n)
{
   for (uint64_t i = 0; i < n; i++)
   {
      volatile float ii = iq[i].x * iq[i].x;
      volatile float qq = iq[i].y * iq[i].y;
      out[i] = __builtin_sqrtf(ii + qq);
   }
}

/* the radio branch of the adaptive resampler (SignalResamplingTask.cpp:168-226), one buffer after the other */
void nfc_resample_radio_kernel(const float *__restrict__ in, uint64_t pitchFloats, uint32_t nBuffers, uint32_t n,
                               float *__restrict__ out, uint64_t outPitchFloats, uint32_t capacityPairs, uint32_t *__restrict__ counts)
{
   const int32_t window = 51, interval = 255;
   const float filter = 0.005f;

   for (uint32_t buffer = 0; buffer < nBuffers; buffer++)
   {
      const float *x = in + (uint64_t)buffer * pitchFloats;
      float *dst = out + (uint64_t)buffer * outPitchFloats;
      uint32_t count = 0;

      auto put = [&](float value, float offset) {
         if (count < capacityPairs)
         {
            dst[2 * count] = value;
            dst[2 * count + 1] = offset;
         }
         count++;
      };

      float avrg = 0.0f;
      for (int32_t k = 0; k < window / 2; k++)
         avrg += x[k];

      float last = x[0];
      put(x[0], 0.0f);

      int32_t c = 0, p = -1;

      for (int32_t i = 0; i < (int32_t)n; ++i, ++p)
      {
         const float value = x[i];

         if ((uint32_t)(i + window / 2) < n)
            avrg += x[i + window / 2];

         if (i - window / 2 - 1 >= 0)
            avrg -= x[i - window / 2 - 1];

         const float stdev = std::fabs(value - (avrg / (float)window));

         if (stdev > filter || (i - c) >= interval)
         {
            if (stdev > filter && c < p)
               put(last, (float)p);

            put(value, (float)i);
            c = i;
         }

         last = value;
      }

      if (c < p)
         put(last, (float)p);

      counts[buffer] = count;
   }
}

/* ------------------------------------------------------------------------------------------ */
/* time-parallel path: the same device functions (nfc_scan.hpp), chunk after chunk / lane after lane */
/* ------------------------------------------------------------------------------------------ */

namespace {

float sample_of(const uint8_t *data, uint32_t stride, uint32_t i)
{
   return emu_sample_at(data, stride, i);
}

void copy_lane(const NfcLaunch &from, uint32_t a, const NfcLaunch &to, uint32_t b)
{
   const float *src = from.rings + (uint64_t)(a / NFC_LANES) * from.ringBlockFloats + (a % NFC_LANES);
   float *dst = to.rings + (uint64_t)(b / NFC_LANES) * to.ringBlockFloats + (b % NFC_LANES);

   for (uint32_t i = 0; i < from.ringBlockFloats / NFC_LANES; i++)
      dst[(uint64_t)i * NFC_LANES] = src[(uint64_t)i * NFC_LANES];

   std::memcpy(to.bytes + (uint64_t)b * NFC_STREAM_BYTES, from.bytes + (uint64_t)a * NFC_STREAM_BYTES, NFC_STREAM_BYTES);
}

}

void nfc_scan_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A)
{
   const uint32_t L = A.params.chunkSamples, WU = A.params.warmSamples;

   for (uint32_t listed = 0; listed < A.nChunks + A.nChunksMore; listed++)
   {
      NfcScanChunk ch = listed < A.nChunks ? A.chunks[listed] : A.chunksMore[listed - A.nChunks];
      const bool repair = (ch.index & NFC_CHUNK_REPAIR) != 0;
      const bool envelopeOnly = repair && (ch.index & NFC_CHUNK_ENVELOPE) != 0;
      ch.index &= ~(NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE);
      const NfcScanJob *job = A.jobs + ch.job;
      const uint32_t g = job->firstChunk + ch.index;
      const uint32_t start = ch.index * L;
      const uint32_t end = start + L < job->count ? start + L : job->count;
      const uint32_t walkFrom = (ch.index == 0 || repair) ? start : start - WU;
      const NfcStreamState *st = A.states + job->slot;

      if (envelopeOnly)
      {
         /* the envelope tracker alone, from the true start, until it meets the first walk's trajectory (nfc_scan_kernel) */
         NfcScanSeam seam = A.seams[g];
         float env = seam.start.env;
         uint32_t pf = seam.start.pulseFilter, clock = st->clock + start;
         float lo = NFC_SCAN_BIG, hi = -NFC_SCAN_BIG;
         bool merged = false;

         if ((start % NFC_SCAN_POINT) == 0)
         {
            NfcScanPoint &first = A.points[job->firstPoint + start / NFC_SCAN_POINT];
            first.env = seam.start.env;
            first.pulseFilter = seam.start.pulseFilter;
         }

         for (uint32_t sp = start; sp < end; sp++)
         {
            if (sp > start && (sp % NFC_SCAN_POINT) == 0)
            {
               NfcScanPoint &stored = A.points[job->firstPoint + sp / NFC_SCAN_POINT];
               if (nfc_bits(stored.env) == nfc_bits(env) && stored.pulseFilter == pf)
               {
                  merged = true;
                  break;
               }
               stored.env = env;
               stored.pulseFilter = pf;
            }

            ++clock;
            ++pf;
            nfc_envelope_step(*cfgPtr, clock, pf, env, sample_of(job->data, A.stride, sp));
            lo = env < lo ? env : lo;
            hi = env > hi ? env : hi;

            if ((sp % NFC_SCAN_TILE) == NFC_SCAN_TILE - 1 || sp == end - 1)
            {
               NfcScanTile &stat = A.tileStats[job->firstTile + sp / NFC_SCAN_TILE];
               stat.envmin = lo;
               stat.envmax = hi;
               stat.bits |= NFC_TILE_REWALKED;
               lo = NFC_SCAN_BIG;
               hi = -NFC_SCAN_BIG;
            }
         }

         if (!merged)
         {
            seam.end.env = env;
            seam.end.pulseFilter = pf;
         }
         A.seams[g] = seam;
         continue;
      }

      NfcScanLane w;
      std::memset(&w, 0, sizeof(w));
      NfcScanSeam seam;
      std::memset(&seam, 0, sizeof(seam));
      bool begun = false;
      bool merged = false;

      for (uint32_t sp = walkFrom; sp < end && !merged; sp++)
      {
         const float x = sample_of(job->data, A.stride, sp);

         if (!begun)
         {
            /* guess for the envelope: mean of the first tile of the walk (nfc_scan_kernel does the same) */
            float first = 0.0f;
            const uint32_t span = end - sp < NFC_SCAN_TILE ? end - sp : NFC_SCAN_TILE;
            for (uint32_t k = 0; k < span; k++)
               first += sample_of(job->data, A.stride, sp + k);
            first = first / (float)span;

            if (repair)
               nfc_scan_resume(w, A.seams[g].start, A.seams[g].start.edgeTime, st->clock + sp);
            else
               nfc_scan_begin(w, ch.index == 0 ? st : nullptr, st->clock + sp, first);
            begun = true;
         }

         if (ch.index != 0 && !repair && sp == walkFrom + WU / 3 / NFC_SCAN_TILE * NFC_SCAN_TILE)
            nfc_scan_reseed(w);

         if (sp == start)
            nfc_scan_point(w, seam.start);

         if (sp >= start && (sp % NFC_SCAN_POINT) == 0)
         {
            NfcScanPoint &stored = A.points[job->firstPoint + sp / NFC_SCAN_POINT];

            /* a second walk that has met the first one's trajectory ends here (nfc_scan_kernel does the same) */
            if (repair && sp > start)
            {
               NfcScanPoint here;
               nfc_scan_point(w, here);

               if (nfc_scan_merged(here, stored))
               {
                  const uint32_t atMerge = stored.edgeTime;
                  for (uint32_t q = sp + NFC_SCAN_POINT; q < end; q += NFC_SCAN_POINT)
                     nfc_scan_adopt(A.points[job->firstPoint + q / NFC_SCAN_POINT], atMerge, w.fe.edgeTime);
                  seam.end = A.seams[g].end;
                  nfc_scan_adopt(seam.end, atMerge, w.fe.edgeTime);
                  stored = here;
                  merged = true;
                  break;
               }
            }

            nfc_scan_point(w, stored);
         }

         nfc_scan_sample(*cfgPtr, w, x);

         if ((sp % NFC_SCAN_TILE) == NFC_SCAN_TILE - 1 || sp == end - 1)
         {
            NfcScanTile stat;
            nfc_scan_tile_end(w, stat);
            if (repair)
               stat.bits |= NFC_TILE_REWALKED;
            if (sp >= start)
               A.tileStats[job->firstTile + sp / NFC_SCAN_TILE] = stat;
         }
      }

      if (begun && !merged)
         nfc_scan_point(w, seam.end);
      if (repair)
         seam.start = A.seams[g].start;
      A.seams[g] = seam;
   }
}

/* the envelope tracker's second walks: the kernel's own text, a chunk after the other */
void nfc_envelope_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A)
{
   for (uint32_t listed = 0; listed < A.nChunks; listed++)
      nfc_envelope_rewalk(*cfgPtr, A, A.chunks[listed]);
}

/* the second walk of the scan (repair form, from the verified chunk starts): front-end planes only */
void nfc_scan_planes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A)
{
   const uint32_t L = A.planesPiece ? A.planesPiece : A.params.chunkSamples;

   const uint32_t per = A.planesPerChunk ? A.planesPerChunk : 1u;

   for (uint32_t listed = 0; listed < A.nChunks * per; listed++)
   {
      NfcScanChunk ch = A.chunks[listed / per];
      ch.index &= ~(NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE);
      if (A.planesPerChunk)
         ch.index = ch.index * per + listed % per; /* (an entry that is a chunk walked a piece per lane: nfc_kernels.hip) */
      const NfcScanJob *job = A.jobs + ch.job;
      const uint32_t start = ch.index * L;
      const uint32_t end = start + L < job->count ? start + L : job->count;
      const NfcStreamState *st = A.states + job->slot;

      if (start >= end)
         continue;

      /* (from the chunk's start, or from the stored point: nfc_kernels.hip) */
      const NfcScanPoint &from = A.planesPiece ? A.points[job->firstPoint + start / NFC_SCAN_POINT] : A.seams[job->firstChunk + ch.index].start;

      NfcScanLane w;
      nfc_scan_resume(w, from, from.edgeTime, st->clock + start);

      float *out = A.planes + 4u * ((uint64_t)job->firstTile * NFC_SCAN_TILE);

      for (uint32_t sp = start; sp < end; sp++)
      {
         const float filtered = nfc_scan_sample(*cfgPtr, w, sample_of(job->data, A.stride, sp));
         out[4u * (uint64_t)sp + 0] = filtered;
         out[4u * (uint64_t)sp + 1] = w.fe.env;
         out[4u * (uint64_t)sp + 2] = w.fe.mdev;
         out[4u * (uint64_t)sp + 3] = w.fe.avg;
      }
   }
}

void nfc_planes_stale_kernel(NfcScanArgs A, const NfcScanChunk *all, uint32_t nAll, NfcScanChunk *out, uint32_t *count)
{
   for (uint32_t g = 0; g < nAll; g++)
      if (A.planesStale[g])
         out[emu_add(count, 1u)] = all[g];
}

void nfc_seams_kernel(NfcScanArgs A, uint32_t first)
{
   for (uint32_t j = 0; j < A.nJobs; j++)
   {
      NfcScanJob job = A.jobs[j];
      if (first)
         job.passes = 0;
      if (!(job.status & NFC_JOB_INVALID))
         nfc_seams_check(job, j, A.seams, A.chunkEdge, A.states[job.slot].edgeTime, A.repairs, A.repairCount, A.points, A.params.chunkSamples, A.repairsEnv, A.repairEnvCount,
                         A.planesStale);
      A.jobs[j] = job;
   }
}

void nfc_tiles_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, uint32_t nTilesTotal)
{
   (void)nTilesTotal;
   for (uint32_t j = 0; j < A.nJobs; j++)
   {
      NfcScanJob &job = A.jobs[j];
      const uint32_t nTiles = (job.count + NFC_SCAN_TILE - 1) / NFC_SCAN_TILE;
      uint32_t rewalked = 0;
      for (uint32_t i = 0; i < nTiles; i++)
      {
         const uint32_t flags = nfc_tile_flags(*cfgPtr, A.params, A.tileStats + job.firstTile, i);
         A.tiles[job.firstTile + i] = flags;
         if (flags & NFC_TILE_OFFGRID)
            job.status |= A.params.offGridAlone ? (NFC_JOB_ALONE | NFC_JOB_OFFGRID_SEEN) : NFC_JOB_OFFGRID;
         if ((flags & NFC_TILE_BUSY) && !(flags & NFC_TILE_DARK))
            job.busyTiles++;
         rewalked += (flags & NFC_TILE_REWALKED) ? 1 : 0;
      }
      if (std::getenv("NFC_EMU_DEBUG"))
         std::fprintf(stderr, "[emu] job %u: %u of %u tiles had their front end walked again\n", j, rewalked, nTiles);
   }
}

/* the kernel's form of the rule (masks of 64 tiles, nfc_group_*), tile by tile where the kernel uses a ballot */
static uint32_t emu_windows_place(const NfcScanJob &job, uint32_t j, const uint32_t *t, uint32_t nTiles, NfcWindow *out, uint32_t room, bool write)
{
   const uint32_t groups = (nTiles + 63u) / 64u;
   NfcWindowPlacer placer {0u, 0u};
   uint64_t busyBefore = ~0ull;

   for (uint32_t g = 0; g < groups; g++)
   {
      uint64_t busy = 0, cluster = 0, cut = 0;

      for (uint32_t l = 0; l < 64u && g * 64u + l < nTiles; l++)
         if (t[g * 64u + l] & NFC_TILE_BUSY)
            busy |= 1ull << l;

      for (uint32_t l = 0; l < 64u && g * 64u + l < nTiles; l++)
      {
         const uint32_t i = g * 64u + l;
         if (nfc_tile_may_start(i, job.count) && nfc_group_cluster(busyBefore, busy, l))
            cluster |= 1ull << l;
         if (!(t[i] & (NFC_TILE_RETIRE_OK | NFC_TILE_DARK)) && nfc_tile_may_cut(i, job.count))
            cut |= 1ull << l;
      }

      if (cluster | cut)
         nfc_group_place(placer, job, j, out, room, g, cluster, cut, write);

      busyBefore = busy;
   }

   nfc_windows_close(placer, job, j, out, room, write);
   return placer.n;
}

static void emu_windows_marks(uint32_t *t, uint32_t nTiles)
{
   const uint32_t groups = (nTiles + 63u) / 64u;
   uint64_t blockedNext = ~0ull;
   uint32_t carry = 0;

   for (uint32_t g = groups; g-- > 0;)
   {
      uint64_t blocked = 0, dark = 0;

      for (uint32_t l = 0; l < 64u; l++)
      {
         const uint32_t i = g * 64u + l;
         if (i >= nTiles || (t[i] & NFC_TILE_BUSY))
            blocked |= 1ull << l;
         if (i < nTiles && (t[i] & NFC_TILE_DARK))
            dark |= 1ull << l;
      }

      const bool full = g * 64u + 64u <= nTiles;

      for (uint32_t l = 0; l < 64u && g * 64u + l < nTiles; l++)
      {
         const uint32_t i = g * 64u + l;
         const uint32_t run = nfc_group_dark_run(dark, full, l, carry);
         const bool ok = nfc_group_retire_ok(blocked, blockedNext, l);
         t[i] = (t[i] & 0xFFFFu & ~NFC_TILE_RETIRE_OK) | (ok ? NFC_TILE_RETIRE_OK : 0u) | (run << NFC_TILE_DARK_RUN_SHIFT);
      }

      carry = nfc_group_dark_run(dark, full, 0u, carry);
      blockedNext = blocked;
   }
}

void nfc_windows_kernel(NfcScanArgs A)
{
   for (uint32_t j = 0; j < A.nJobs; j++)
   {
      NfcScanJob job = A.jobs[j];

      job.status &= ~NFC_JOB_OVERFLOW;

      if (job.status & NFC_JOB_INVALID)
      {
         job.windows = 0;
         A.jobs[j] = job;
         continue;
      }

      const uint32_t nTiles = (job.count + NFC_SCAN_TILE - 1) / NFC_SCAN_TILE;

      /* the rule as stated (nfc_windows_build), on a copy, to hold the kernel's form against */
      std::vector<uint32_t> stated(A.tiles + job.firstTile, A.tiles + job.firstTile + nTiles);
      const uint32_t statedNeed = nfc_windows_build(job, j, stated.data() - job.firstTile, nullptr, 0);

      emu_windows_marks(A.tiles + job.firstTile, nTiles);

      const bool solo = job.count <= A.params.soloSamples || (job.status & NFC_JOB_ALONE) != 0u; /* (its carry lane alone, nfc_windows_kernel) */
      const uint32_t placed = emu_windows_place(job, j, A.tiles + job.firstTile, nTiles, nullptr, 0, false);
      const uint32_t need = solo ? 0u : placed;
      const uint32_t first = emu_add(A.windowCount, need);

      if (placed != statedNeed || std::memcmp(stated.data(), A.tiles + job.firstTile, 4u * nTiles) != 0)
      {
         std::fprintf(stderr, "[emu] window rule: the mask form differs from nfc_windows_build (job %u: %u vs %u windows)\n", j, need, statedNeed);
         std::abort();
      }

      job.firstWindow = A.firstWindowSlot + first;
      job.windows = need;

      if (first + need > A.windowRoom)
      {
         job.status |= NFC_JOB_OVERFLOW;
         job.windows = 0;
      }
      else if (need)
      {
         emu_windows_place(job, j, A.tiles + job.firstTile, nTiles, A.windows + job.firstWindow, need, true);

         std::vector<NfcWindow> want(need);
         nfc_windows_build(job, j, stated.data() - job.firstTile, want.data(), need);
         if (need && std::memcmp(want.data(), A.windows + job.firstWindow, sizeof(NfcWindow) * need) != 0)
         {
            std::fprintf(stderr, "[emu] window rule: the mask form places windows differently from nfc_windows_build (job %u)\n", j);
            std::abort();
         }
      }

      A.jobs[j] = job;

      if (std::getenv("NFC_EMU_DEBUG"))
      {
         uint32_t kinds[8] = {0};
         uint32_t busy = 0, ok = 0;
         for (uint32_t i = 0; i < nTiles; i++)
         {
            const uint32_t f = A.tiles[job.firstTile + i];
            for (int b = 0; b < 6; b++)
               kinds[b] += (f >> b) & 1u;
            busy += (f & NFC_TILE_BUSY) ? 1 : 0;
            ok += (f & NFC_TILE_RETIRE_OK) ? 1 : 0;
         }
         std::fprintf(stderr, "[emu] job %u slot %u count %u tiles %u busy %u (range %u edge %u unarmed %u carrier %u offgrid %u nohist %u) retireOK %u status %x windows %u\n",
                      j, job.slot, job.count, nTiles, busy, kinds[0], kinds[1], kinds[2], kinds[3], kinds[4], kinds[5], ok, job.status, job.windows);
         for (uint32_t i = 0; i < job.windows && i < 40; i++)
            std::fprintf(stderr, "[emu]   window %u start %u activate %u\n", job.firstWindow + i, A.windows[job.firstWindow + i].start, A.windows[job.firstWindow + i].activate);
      }
   }
}

void nfc_carry_lanes_kernel(NfcScanArgs A, NfcLaunch real, NfcLaunch lanes, uint32_t pass)
{
   for (uint32_t j = 0; j < A.nJobs; j++)
   {
      const NfcScanJob *job = A.jobs + j;
      const uint32_t noHand = pass ? A.windows[j].noHand : 0u;

      if (pass && !A.windows[j].rerun)
      {
         A.works[j].count = 0;
         continue;
      }

      copy_lane(real, job->slot, lanes, j);

      NfcStreamState s = real.states[job->slot];
      NfcStreamCold cold = real.cold[job->slot];
      cold.frameHead = 0;
      cold.frameTail = 0;
      cold.emitOwn = 0; /* (a lane's notes: NfcStreamCold) */
      cold.waitFlags = 0;
      cold.waitUsed[0] = cold.waitUsed[1] = cold.waitUsed[2] = cold.waitUsed[3] = 0;
      lanes.states[j] = s;
      lanes.cold[j] = cold;

      NfcWindow w;
      std::memset(&w, 0, sizeof(w));
      w.job = j;
      w.verify = 0xFFFFFFFFu;
      w.noHand = noHand;
      nfc_carry_take(w.carry, s, cold);
      w.want = w.carry;
      A.windows[j] = w;

      NfcWork work;
      work.data = job->data;
      work.count = (job->status & NFC_JOB_INVALID) ? 0u : job->count;
      work.stride = A.stride;
      work.tiles = A.tiles + job->firstTile;
      A.works[j] = work;
   }
}

void nfc_window_lanes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, NfcLaunch lanes, uint32_t pass, uint32_t order, uint32_t lenLo, uint32_t lenHi)
{
   const uint32_t n = *A.windowCount < A.windowRoom ? *A.windowCount : A.windowRoom;
   const uint32_t wEnd = A.firstWindowSlot + n;

   for (uint32_t wi = A.firstWindowSlot; wi < wEnd; wi++)
   {
      NfcWindow w = A.windows[wi];
      const NfcScanJob *job = A.jobs + w.job;

      /* (the order of the run list: nfc_kernels.hip) */
      bool listed = true;
      if (order != 0u)
      {
         const uint32_t next = (wi + 1u < wEnd && A.windows[wi + 1u].job == w.job) ? A.windows[wi + 1u].start : job->count;
         listed = next - w.start >= lenLo && next - w.start < lenHi;
      }

      if (order == 2u)
      {
         if (listed && A.works[wi].count != 0u)
            A.runList[emu_add(A.runCount, 1u)] = wi;
         continue;
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

         const uint32_t chunk = job->firstChunk + w.start / A.params.chunkSamples;

         NfcStreamState s;
         NfcStreamCold cold;
         {
            const NfcScanPoint &pt = A.points[job->firstPoint + w.start / NFC_SCAN_POINT];
            w.tracked = (pt.zone & NFC_ZONE_EDGE_KNOWN) ? pt.edgeTime : A.chunkEdge[chunk];
            nfc_window_lane(*cfgPtr, w, pt, A.states[job->slot].clock, s, cold);
         }

         lanes.states[wi] = s;
         lanes.cold[wi] = cold;
         A.windows[wi] = w;

         work.count = job->count - w.start;

         if (listed)
            A.runList[emu_add(A.runCount, 1u)] = wi;
      }

      A.works[wi] = work;
   }
}

void nfc_final_lanes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, NfcLaunch lanes)
{
   for (uint32_t j = 0; j < A.nJobs; j++)
   {
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
         const uint32_t chunk = job->firstChunk + w.start / A.params.chunkSamples;

         NfcStreamState s;
         NfcStreamCold cold;
         {
            const NfcScanPoint &pt = A.points[job->firstPoint + w.start / NFC_SCAN_POINT];
            w.tracked = (pt.zone & NFC_ZONE_EDGE_KNOWN) ? pt.edgeTime : A.chunkEdge[chunk];
            nfc_window_lane(*cfgPtr, w, pt, A.states[job->slot].clock, s, cold);
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
}

void nfc_chain_kernel(NfcScanArgs A, NfcLaunch lanes, uint32_t maxPasses)
{
   for (uint32_t j = 0; j < A.nJobs; j++)
   {
      NfcScanJob job = A.jobs[j];

      if (job.status & NFC_JOB_INVALID)
         continue;

      if (job.passes > 0 && !(job.status & NFC_JOB_RERUN))
         continue;

      const bool again = nfc_chain_follow(job, j, A.windows, lanes.states, lanes.cold, maxPasses);
      if (again)
         emu_add(A.rerunCount, 1u);

      A.jobs[j] = job;

      if (std::getenv("NFC_EMU_DEBUG"))
      {
         uint64_t steps = 0, liveSteps = 0;
         for (uint32_t i = 0; i <= job.windows; i++)
         {
            const uint32_t lane = i == 0 ? j : job.firstWindow + i - 1;
            const NfcWindow &w = A.windows[lane];
            steps += w.stop - w.start;
            liveSteps += w.live ? w.stop - w.start : 0;
         }
         std::fprintf(stderr, "[emu] chain job %u pass %u again %d final %u status %x lanes %u lane-steps %llu (live %llu) of %u samples\n", j, job.passes, (int)again,
                      job.finalLane, job.status, job.windows + 1, (unsigned long long)steps, (unsigned long long)liveSteps, job.count);
         for (uint32_t i = 0; i <= job.windows && i < 60; i++)
         {
            const uint32_t lane = i == 0 ? j : job.firstWindow + i - 1;
            const NfcWindow &w = A.windows[lane];
            std::fprintf(stderr, "[emu]   lane %u start %u act %u stop %u retired %u live %u rerun %u frames %u\n", lane, w.start, w.activate, w.stop, w.retired,
                         w.live, w.rerun, lanes.cold[lane].framesOut);
         }
      }
   }
}

void nfc_finish_kernel(NfcScanArgs A, NfcLaunch real, NfcLaunch lanes)
{
   if (lanes.sinkCtl[1])
      emu_add(real.sinkCtl + 1, lanes.sinkCtl[1]);

   for (uint32_t j = 0; j < A.nJobs; j++)
   {
      const NfcScanJob *job = A.jobs + j;

      if (job->status & NFC_JOB_INVALID)
         continue;

      nfc_finish_frames(*job, j, A.windows, lanes.cold, lanes.sink, real.sink, real.sinkCtl, real.sinkWords);

      const uint32_t saved = job->finalLane == j ? 0u : A.windows[job->finalLane].saved;
      const uint32_t from = job->finalLane == j ? j : (saved ? job->finalLane : A.finalLaneSlot + j);

      if (saved)
      {
         const uint32_t rows = real.ringBlockFloats / NFC_LANES;
         float *dst = real.rings + (uint64_t)(job->slot / NFC_LANES) * real.ringBlockFloats + (job->slot % NFC_LANES);
         for (uint32_t i = 0; i < rows; i++)
            dst[(uint64_t)i * NFC_LANES] = A.saveRings[(uint64_t)(saved - 1u) * rows + i];
         std::memcpy(real.bytes + (uint64_t)job->slot * NFC_STREAM_BYTES, A.saveBytes + (uint64_t)(saved - 1u) * NFC_STREAM_BYTES, NFC_STREAM_BYTES);
      }
      else
         copy_lane(lanes, from, real, job->slot);

      NfcStreamState s = lanes.states[from];
      NfcStreamCold cold = lanes.cold[from];
      if (job->finalLane != j)
         nfc_final_fixup(s, cold, A.windows[job->finalLane]);
      cold.frameHead = 0;
      cold.frameTail = 0;
      std::memset(cold.boundF, 0, sizeof(cold.boundF));
      std::memset(cold.waitUsed, 0, sizeof(cold.waitUsed));
      cold.waitFlags = 0;
      cold.emitOwn = 0;
      cold.trackedEnd = 0;
      real.states[job->slot] = s;
      real.cold[job->slot] = cold;
   }
}

/* the wave decoder: tests/hostsim/emu_wave.cpp runs the kernel's own text, 64 fibres per wave */
void emu_wave_kernel(const NfcConfig *cfgPtr, const NfcLaunch &L, const NfcScanArgs &A, uint32_t mode, uint32_t blocks);

void nfc_wave_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, NfcScanArgs A, uint32_t mode)
{
   emu_wave_kernel(cfgPtr, L, A, mode, fakehip::launchGrid.x);
}

void nfc_read_kernel(const float4 *__restrict__ data, uint64_t n, float *__restrict__ out)
{
   float acc = 0.0f;
   for (uint64_t i = 0; i < n; i++)
      acc += data[i].x;
   if (acc == 12345.678f)
      out[0] = acc;
}

/* (test hook: the device the calling thread's last C-ABI call left current) */
extern "C" int nfcgpu_emulated_current_device()
{
   return fakehip::currentDevice;
}
