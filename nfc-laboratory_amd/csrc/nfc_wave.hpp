/*
 * nfc_wave.hpp — the wave decoder: one wavefront decodes one lane of work of the time-parallel path (nfc_scan.h: the
 * carry lane of a stream, a speculative window, the lane that regenerates a stream's final state), 64 consecutive
 * samples per step, one sample per lane.
 *
 * Why. The stream-parallel step machine (nfc_core.hpp: one lane per stream, rings in HBM) pays ~1000 instruction slots
 * and a round trip to its history rings per sample; a lane of the time-parallel path is one stream, so a pass costs its
 * longest lane at 2-3 us per sample. Here the wave is the lane:
 *
 *   - the front end (NfcTech.cpp:28-105) is not computed at all: the scan kernel's second walk has left its results per
 *     sample ({filtered, envelope, deviation, average}: NfcScanArgs::planes), 64 of them are one coalesced 1 KiB load;
 *   - one stream's history and correlation rings live in LDS (NfcWaveLds, ~21 KiB), the history 1024 deep so that a
 *     tile is written ahead of the sample at hand;
 *   - the decoder's state is wave-uniform (every lane holds the same copy); a sample that can change it is handled by
 *     the very step of nfc_core.hpp (nfc_step_impl<.., GIVEN>) executed uniformly by the wave, so whatever the
 *     reference does at that sample (NfcDecoder.cpp:394-418 and the tech decoders) is done by the same statement the
 *     stream-parallel kernels run;
 *   - samples that provably change nothing but the running sums and ring entries are not stepped: the wave evaluates
 *     the gate of the mode at hand for all remaining samples of the tile at once (nfc_wave_fast_*: correlations from
 *     wave prefix sums over the tile plus the ring as the tile found it, bit for bit the values the step would form),
 *     commits sums and ring entries of the uneventful run in bulk and steps the first eventful sample.
 *
 * Included by nfc_wave.hip (HIP, gfx950) and by tests/hostsim/emu_wave.cpp (test infrastructure: the same text run by 64
 * fibres per wave on the CPU). The includer provides, besides what nfc_core.hpp and nfc_scan.hpp ask for:
 *   NFC_WAVE_LANE()               lane 0..63
 *   NFC_WAVE_BARRIER()            LDS written before it is visible to every lane after it
 *   NFC_WAVE_BALLOT(p)            uint64_t
 *   NFC_WAVE_UNIFORM_BEGIN(u) / NFC_WAVE_UNIFORM_END(u)
 *                                 bracket code every lane executes identically on the uniform record `u` (the device
 *                                 runs it in all lanes; the fibre build runs it in lane 0 and copies `u` to the others)
 *   NFC_WAVE_LDS                  address space qualifier of LDS objects
 */
#ifndef NFC_AMD_WAVE_HPP
#define NFC_AMD_WAVE_HPP

#include "nfc_scan_launch.h"

#if NFC_HIST != 1024u || NFC_RING_STRIDE != 1u
#error "nfc_wave.hpp needs nfc_core.hpp compiled with NFC_HIST 1024 and NFC_RING_STRIDE 1"
#endif

#define NFC_WAVE_CARRY 0u   /* work items = jobs: the lane that continues a stream from its own state */
#define NFC_WAVE_WINDOWS 1u /* work items = entries of the run list: speculative windows */
#define NFC_WAVE_FINAL 2u   /* work items = jobs: the lane that regenerates the state of a stream's last window */

#define NFC_WAVE_RING_FLOATS (4u * NFC_HIST + NFC_PROD + NFC_CORR_MAX)

/* LDS of one wave */
struct NfcWaveLds
{
   float ring[NFC_WAVE_RING_FLOATS]; /* the regions of nfc_core.hpp (NFC_R_*), one stream */
   NfcStreamCold cold;
   uint8_t bytes[NFC_STREAM_BYTES];
   uint32_t flags;                   /* NfcStreamCold::usedTech while the lane runs */
   float env[NFC_LANES];             /* envelope / average after each sample of the tile at hand */
   float avg[NFC_LANES];
   float scratch[NFC_LANES];
   float sum[7][NFC_LANES];          /* running sums after each sample of the tile (nfc_wave_fast.hpp); [6]: hand-over to the uniform part */
};

/* what the lanes of the wave hold identically */
struct NfcWaveUni
{
   NfcStreamState s;
   uint32_t consumed; /* samples of the lane's row taken */
   uint32_t stepped;  /* samples stepped one by one (statistics) */
   uint32_t stopped;  /* 1 retired at rest, 2 handed over */
   uint32_t succ;
   uint32_t at;       /* sample of the tile at hand */
};

NFC_DEV bool nfc_wave_exact_span(uint32_t clock, uint32_t count)
{
   const uint32_t start = clock + 1u + 1024u;
   const uint32_t untilWrap = 0u - start;
   return count != 0 && (start < 2048u || untilWrap < count);
}

/* what a lane works on */
struct NfcWaveItem
{
   uint32_t w;         /* lane slot: index into L.states / L.cold / L.windows / L.works */
   uint32_t mode;
   const uint8_t *data;
   uint32_t count;     /* samples of the row */
   const uint32_t *tiles;
   uint32_t startPos;  /* stream position of the row's first sample */
   uint32_t clockBase; /* clock of the sample before the submission */
   const float *planes; /* record of stream position 0 */
   const NfcScanJob *job;
};

/* the decoder's edge time after the sample at stream position `last`: the edge-peak tracker (NfcTech.cpp:86-104) walked
 * from the stored point at or before it over the filtered plane, then what the decoder's own copy holds (zeroed by the
 * last carrier frame unless the tracker has moved since: nfc_edge_time). Called by every lane. */
NFC_DEV uint32_t nfc_wave_edge_time(const NfcConfig &c, const NfcScanArgs &A, const NfcWaveItem &it, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t last, bool emitValid, uint32_t emitClock)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t q = last / NFC_SCAN_POINT;
   const NfcScanPoint &pt = A.points[it.job->firstPoint + q];

   uint32_t tracked = (pt.zone & NFC_ZONE_EDGE_KNOWN) ? pt.edgeTime : A.chunkEdge[it.job->firstChunk + (q * NFC_SCAN_POINT) / A.params.chunkSamples];
   float peak = pt.edgePeak;

   for (uint32_t base = q * NFC_SCAN_POINT; base <= last; base += NFC_LANES)
   {
      const uint32_t n = last - base + 1u < NFC_LANES ? last - base + 1u : NFC_LANES;

      NFC_WAVE_BARRIER();
      lds->scratch[lane] = lane < n ? nfc_abs(it.planes[4u * (uint64_t)(base + lane)]) : 0.0f;
      NFC_WAVE_BARRIER();

      for (uint32_t k = 0; k < n; k++)
      {
         const float rectified = lds->scratch[k];
         const uint32_t clock = it.clockBase + 1u + base + k;
         const bool high = rectified > c.highThreshold;
         const bool top = high && rectified > peak;
         const bool low = !high && rectified < c.lowThreshold;

         tracked = top ? clock : tracked;
         peak = top ? rectified : (low ? 0.0f : peak);
      }
   }

   return (emitValid && (int32_t)(tracked - emitClock) <= 0) ? 0u : tracked;
}

/* a stream's rings between HBM ([slot][64 lanes], history NFC_HIST_STORED deep) and LDS (history NFC_HIST deep): the
 * stored history holds the last NFC_HIST_STORED samples up to `clock` */
NFC_DEV void nfc_wave_rings_in(NFC_WAVE_LDS NfcWaveLds *lds, const float *src, uint64_t pitch, uint32_t clock, uint32_t corrTotal)
{
   const uint32_t lane = NFC_WAVE_LANE();

   for (uint32_t i = lane; i < NFC_WAVE_RING_FLOATS; i += NFC_LANES)
      lds->ring[i] = 0.0f;

   NFC_WAVE_BARRIER();

   for (uint32_t r = 0; r < 4u; r++)
   {
      for (uint32_t k = lane; k < NFC_HIST_STORED; k += NFC_LANES)
      {
         const uint32_t t = clock - k; /* sample clock */
         lds->ring[r * NFC_HIST + (t & (NFC_HIST - 1u))] = src[(uint64_t)(r * NFC_HIST_STORED + (t & (NFC_HIST_STORED - 1u))) * pitch];
      }
   }

   for (uint32_t k = lane; k < NFC_PROD + corrTotal; k += NFC_LANES)
      lds->ring[4u * NFC_HIST + k] = src[(uint64_t)(4u * NFC_HIST_STORED + k) * pitch];

   NFC_WAVE_BARRIER();
}

NFC_DEV void nfc_wave_rings_out(const NFC_WAVE_LDS NfcWaveLds *lds, float *dst, uint64_t pitch, uint32_t clock, uint32_t corrTotal)
{
   const uint32_t lane = NFC_WAVE_LANE();

   NFC_WAVE_BARRIER();

   for (uint32_t r = 0; r < 4u; r++)
   {
      for (uint32_t k = lane; k < NFC_HIST_STORED; k += NFC_LANES)
      {
         const uint32_t t = clock - k;
         dst[(uint64_t)(r * NFC_HIST_STORED + (t & (NFC_HIST_STORED - 1u))) * pitch] = lds->ring[r * NFC_HIST + (t & (NFC_HIST - 1u))];
      }
   }

   for (uint32_t k = lane; k < NFC_PROD + corrTotal; k += NFC_LANES)
      dst[(uint64_t)(4u * NFC_HIST_STORED + k) * pitch] = lds->ring[4u * NFC_HIST + k];
}

/* the tile at hand: this lane's sample and the front end's results for it, parked where the step reads them */
struct NfcWaveTile
{
   float x, filt, env, mdev, avg, depth;
};

NFC_DEV NfcWaveTile nfc_wave_load_tile(const NfcWaveItem &it, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t consumed, uint32_t n, uint32_t clock, uint32_t stride)
{
   const uint32_t lane = NFC_WAVE_LANE();
   NfcWaveTile t;
   t.x = t.filt = t.mdev = t.avg = t.depth = 0.0f;
   t.env = 1.0f;

   if (lane < n)
   {
      const uint32_t i = consumed + lane;
      const float *p = it.planes + 4u * (uint64_t)(it.startPos + i);

      t.x = NFC_SAMPLE_AT(it.data, stride, i);
      t.filt = p[0];
      t.env = p[1];
      t.mdev = p[2];
      t.avg = p[3];

      /* modulation depth as the front end forms it (NfcTech.cpp:79-83) */
      const float clamped = (t.x < 0.0f) ? 0.0f : ((t.env < t.x) ? t.env : t.x);
      t.depth = (t.env - clamped) / t.env;

      const uint32_t slot = (clock + 1u + lane) & NFC_HMASK;

      lds->ring[NFC_R_X + slot] = t.x;
      lds->ring[NFC_R_FILT + slot] = t.filt;
      lds->ring[NFC_R_MDEV + slot] = t.mdev;
      lds->ring[NFC_R_DEPTH + slot] = t.depth;
      lds->env[lane] = t.env;
      lds->avg[lane] = t.avg;
   }

   return t;
}

/* ring positions after `n` more samples (the incremental form: nfc_bump n times) */
NFC_DEV void nfc_wave_advance(const NfcConfig &c, NfcStreamState &s, uint32_t n)
{
   s.posA[0] = (s.posA[0] + n) % c.a[0].p1;
   s.posA[1] = (s.posA[1] + n) % c.a[1].p1;
   s.posA[2] = (s.posA[2] + n) % c.a[2].p1;
   s.posF[0] = (s.posF[0] + n) % c.f[1].p1;
   s.posF[1] = (s.posF[1] + n) % c.f[2].p1;
   s.posV1 = (s.posV1 + n) % c.v.p1;
   s.posV0 = (s.posV0 + n) % c.v.p0;
}

/* statistics of the fibre build (tests/hostsim): samples committed in bulk (0) / stepped (1) per stage */
#ifndef NFC_WAVE_COUNT
#define NFC_WAVE_COUNT(key, which, count) ((void)0)
#endif

#include "nfc_wave_fast.hpp"

/* One tile: the next n samples of the lane's row (stream position pos on). allowFast: take the bulk paths (the fibre
 * build runs every tile a second time without them and compares: tests/hostsim/emu_wave.cpp). */
NFC_DEV void nfc_wave_tile(const NfcConfig &cc, const NfcScanArgs &A, const NfcWaveItem &it, NFC_WAVE_LDS NfcWaveLds *lds, const NfcLaneMem &mem, NfcWaveUni &u,
                           NfcWaveFast &fast, uint32_t n, uint32_t pos, bool carry, uint32_t warmFront, uint32_t warm, uint32_t stride, bool allowFast)
{
   const uint32_t lane = NFC_WAVE_LANE();

   NFC_WAVE_BARRIER();
   const NfcWaveTile tile = nfc_wave_load_tile(it, lds, u.consumed, n, u.s.clock, stride);
   NFC_WAVE_BARRIER();

   const bool exact = carry && nfc_wave_exact_span(u.s.clock, n);

   /* per tile: the values of the bulk paths belong to the tile; is the tile on the grid? */
   fast.key = NFC_FK_NONE;
   fast.from = 0;
   fast.clock0 = u.s.clock;
   {
      const float scaled = tile.x * 32768.0f;
      const bool off = lane < n && !(scaled == __builtin_floorf(scaled) && tile.x >= -1.0f && tile.x <= 1.0f);
      if (NFC_WAVE_BALLOT(off))
         fast.gridSince = u.s.clock + n;
   }

   if (u.consumed < warmFront)
   {
      /* history only (nfc_step_front) */
      NFC_WAVE_UNIFORM_BEGIN(u)
      {
         u.s.clock += n;
         nfc_wave_advance(cc, u.s, n);
         u.s.env = lds->env[n - 1u];
         u.s.avg = lds->avg[n - 1u];
         u.s.mdev = lds->ring[NFC_R_MDEV + (u.s.clock & NFC_HMASK)];
      }
      NFC_WAVE_UNIFORM_END(u)
   }
   else
   {
      const bool upkeep = u.consumed < warm;

      u.at = 0;

      while (u.at < n)
      {
         /* samples from u.at on that change nothing but sums and rings: committed in bulk */
         if (!exact && allowFast)
         {
            const uint32_t run = nfc_wave_fast(cc, u, mem, lds, fast, tile, n, upkeep, it);

            if (run)
               continue;
         }

         /* carrier frame due on this sample (NfcDecoder.cpp:472-523)? it is stamped with the decoder's edge time */
         const float avgAt = lds->avg[u.at];
         const bool emits = !upkeep && u.s.lockTech == 0 &&
                            ((avgAt > cc.highThreshold) ? !u.s.carrierOn : ((avgAt < cc.lowThreshold) && !u.s.carrierOff));
         uint32_t edge = 0;

         if (emits)
            edge = nfc_wave_edge_time(cc, A, it, lds, pos + u.at, lds->cold.emitValid != 0, lds->cold.emitClock);

         NFC_WAVE_COUNT(nfc_wave_stage(cc, u.s, upkeep), 1u, 1u);

         NFC_WAVE_UNIFORM_BEGIN(u)
         {
            const uint32_t slot = (u.s.clock + 1u) & NFC_HMASK;

            NfcGiven g;
            g.now.x = lds->ring[NFC_R_X + slot];
            g.now.filt = lds->ring[NFC_R_FILT + slot];
            g.now.mdev = lds->ring[NFC_R_MDEV + slot];
            g.now.depth = lds->ring[NFC_R_DEPTH + slot];
            g.env = lds->env[u.at];
            g.avg = lds->avg[u.at];

            if (emits)
               u.s.edgeTime = edge;

            if (upkeep)
               nfc_step_upkeep<false, true>(cc, u.s, mem, g.now.x, &g);
            else if (exact)
               nfc_step_impl<true, true>(cc, u.s, mem, g.now.x, &g);
            else
               nfc_step_impl<false, true>(cc, u.s, mem, g.now.x, &g);

            u.at++;
            u.stepped++;
         }
         NFC_WAVE_UNIFORM_END(u)
      }
   }

}

/* One lane of work. `lds`: this wave's LDS. Called by all 64 lanes. */
NFC_DEV void nfc_wave_run(const NfcConfig *cfgPtr, const NfcConfig &cc, const NfcLaunch &L, const NfcScanArgs &A, uint32_t mode, uint32_t item,
                          NFC_WAVE_LDS NfcWaveLds *lds)
{
   const uint32_t lane = NFC_WAVE_LANE();

   NfcWaveItem it;
   it.mode = mode;

   if (mode == NFC_WAVE_WINDOWS)
   {
      if (item >= *A.runCount)
         return;
      it.w = A.runList[item];
   }
   else if (mode == NFC_WAVE_FINAL)
      it.w = A.finalLaneSlot + item;
   else
      it.w = item;

   {
      const NfcWork work = L.works[it.w];
      it.data = work.data;
      it.count = work.count;
      it.tiles = work.tiles;
   }

   if (it.count == 0)
      return;

   const bool carry = mode == NFC_WAVE_CARRY;
   const uint32_t stride = L.uniformStride;

   NfcWindow *me = L.windows + it.w;
   it.job = L.jobs + me->job;
   it.startPos = me->start;
   it.clockBase = A.states[it.job->slot].clock;
   it.planes = A.planes + 4u * (uint64_t)it.job->firstTile * NFC_SCAN_TILE;

   const uint32_t verifyPos = me->verify;
   const uint32_t succEnd = it.job->firstWindow + it.job->windows;
   const uint32_t activate = me->activate;
   const uint32_t warmFront = carry ? 0u : NFC_WINDOW_WARM_FRONT;
   const uint32_t warm = carry ? 0u : NFC_WINDOW_WARM_FRONT + NFC_WINDOW_WARM_CORR;

   NfcWaveUni u;
   u.s = L.states[it.w];
   u.consumed = 0;
   u.stepped = 0;
   u.stopped = 0;
   u.at = 0;
   u.succ = carry ? it.job->firstWindow : it.w + 1u;
   if (mode == NFC_WAVE_FINAL || (mode == NFC_WAVE_WINDOWS && (it.w < it.job->firstWindow || it.w >= succEnd)))
      u.succ = succEnd; /* runs on its own */

   /* the front-end recurrences are not walked here (planes): what of them a lane's records are compared by is kept in
    * one form by every lane (nfc_lane_digest); a lane that reaches the end of the submission leaves the scanned state */
   u.s.n1 = 0.0f;
   u.s.edgePeak = 0.0f;
   u.s.pulseFilter = 0u;

   /* LDS: the stream's rings (carry lanes: from the lane's copy of the stream's storage), protocol state, frame bytes */
   const uint64_t pitch = NFC_LANES;
   float *laneRings = L.rings + (uint64_t)(it.w / NFC_LANES) * L.ringBlockFloats + (it.w % NFC_LANES);

   if (carry)
      nfc_wave_rings_in(lds, laneRings, pitch, u.s.clock, cc.corrTotal);
   else
   {
      for (uint32_t i = lane; i < NFC_WAVE_RING_FLOATS; i += NFC_LANES)
         lds->ring[i] = 0.0f;
   }

   for (uint32_t i = lane; i < sizeof(NfcStreamCold) / 4u; i += NFC_LANES)
      ((NFC_WAVE_LDS uint32_t *)&lds->cold)[i] = ((const uint32_t *)(L.cold + it.w))[i];

   for (uint32_t i = lane; i < NFC_STREAM_BYTES / 4u; i += NFC_LANES)
      ((NFC_WAVE_LDS uint32_t *)lds->bytes)[i] = carry ? ((const uint32_t *)(L.bytes + (uint64_t)it.w * NFC_STREAM_BYTES))[i] : 0u;

   if (lane == 0)
      lds->flags = 0u;

   NFC_WAVE_BARRIER();

   NfcLaneMem mem;
   mem.ring = (NFC_RING_FLOAT *)lds->ring;
   mem.lane = 0;
   mem.exact = false;
   mem.linked = true;
   mem.flags = (uint32_t *)&lds->flags;
   mem.bytes = (uint8_t *)lds->bytes;
   mem.sink = L.sink;
   mem.sinkCursor = L.sinkCtl;
   mem.sinkDropped = L.sinkCtl + 1;
   mem.sinkWords = L.sinkWords;
   mem.streamId = it.w;
   mem.cold = (NfcStreamCold *)&lds->cold;
   mem.tables = cfgPtr;

   NfcWaveFast fast;
   nfc_wave_fast_begin(fast);

   /* samples known to be on the capture grid (nfc_wave_fast.hpp): the job's own are (the scan has looked at every one;
    * what it does not check, |x| <= 1, is checked per tile); what a carry lane finds in the stream's rings is not known */
   fast.gridSince = carry ? u.s.clock : u.s.clock - 4096u;
   fast.gridValid = 1;

   for (;;)
   {
      if (u.consumed >= it.count)
         break;

      const uint32_t pos = it.startPos + u.consumed;

      /* ---- tile boundary: publish, retire, hand over (nfc_window_body) ---- */
      const bool wantEdge = pos == verifyPos && pos > 0; /* a published state carries the decoder's edge time */
      uint32_t edgeNow = 0;

      if (wantEdge)
         edgeNow = nfc_wave_edge_time(cc, A, it, lds, pos - 1u, lds->cold.emitValid != 0, lds->cold.emitClock);

      NFC_WAVE_UNIFORM_BEGIN(u)
      {
         if (wantEdge)
            u.s.edgeTime = edgeNow;

         if (pos == verifyPos)
            nfc_lane_publish(*me, u.s, *mem.cold);

         if (u.consumed >= warm && u.consumed > 0 && (it.tiles[u.consumed / NFC_SCAN_TILE] & NFC_TILE_RETIRE_OK) && nfc_quiescent(u.s) &&
             u.s.bankClock == u.s.clock && (uint32_t)(u.s.clock - mem.cold->bankRun) >= NFC_WINDOW_SETTLE)
            u.stopped = 1;

         if (!u.stopped && u.consumed >= warm && u.consumed > 0 && nfc_lane_handover(L.windows, *me, u.succ, succEnd, pos, u.s, *mem.cold))
            u.stopped = 2;
      }
      NFC_WAVE_UNIFORM_END(u)

      if (u.stopped)
         break;

      /* ---- the tile ---- */
      const uint32_t left = it.count - u.consumed;
      const uint32_t n = left < NFC_LANES ? left : NFC_LANES;

#ifdef NFC_WAVE_TILE_HOOK
      NFC_WAVE_TILE_HOOK(cc, A, it, lds, mem, u, fast, n, pos, carry, warmFront, warm, stride);
#else
      nfc_wave_tile(cc, A, it, lds, mem, u, fast, n, pos, carry, warmFront, warm, stride, true);
#endif

      u.consumed += n;
   }

   /* ---- the lane's result ---- */
   const bool ranOut = u.consumed >= it.count;
   const bool atEnd = it.startPos + u.consumed >= it.job->count;
   const bool closing = activate >= it.startPos + it.count;

   uint32_t edgeEnd = 0;
   if (!atEnd && it.startPos + u.consumed > 0)
      edgeEnd = nfc_wave_edge_time(cc, A, it, lds, it.startPos + u.consumed - 1u, lds->cold.emitValid != 0, lds->cold.emitClock);

   NFC_WAVE_UNIFORM_BEGIN(u)
   {
      if (atEnd)
      {
         /* the front end where the submission ends, as the scan left it */
         const uint32_t lastChunk = it.job->firstChunk + it.job->chunks - 1u;
         const NfcScanPoint &p = A.seams[lastChunk].end;
         const uint32_t tracked = (p.zone & NFC_ZONE_EDGE_KNOWN) ? p.edgeTime : A.chunkEdge[lastChunk];

         u.s.env = p.env;
         u.s.n1 = p.n1;
         u.s.mdev = p.mdev;
         u.s.avg = p.avg;
         u.s.edgePeak = p.edgePeak;
         u.s.pulseFilter = p.pulseFilter;
         u.s.edgeTime = (mem.cold->emitValid && (int32_t)(tracked - mem.cold->emitClock) <= 0) ? 0u : tracked;
      }
      else if (it.startPos + u.consumed > 0)
         u.s.edgeTime = edgeEnd;
   }
   NFC_WAVE_UNIFORM_END(u)

   /* 2 handed over, 1 stopped at rest - or out of samples in a state the closing window can take over from -, 0 ran to
    * the end of the submission (nfc_window_body) */
   const uint32_t how = u.stopped == 2 ? 2u : ((!ranOut || (!closing && nfc_lane_comparable(u.s, *mem.cold))) ? 1u : 0u);

   NFC_WAVE_BARRIER();

   if (lane == 0)
      lds->cold.usedTech = lds->flags;

   NFC_WAVE_BARRIER();

   for (uint32_t i = lane; i < sizeof(NfcStreamCold) / 4u; i += NFC_LANES)
      ((uint32_t *)(L.cold + it.w))[i] = ((const NFC_WAVE_LDS uint32_t *)&lds->cold)[i];

   /* rings and frame bytes: a carry or final lane owns storage; a window that ran to the end of the submission with
    * nobody to take over may be the stream's last lane and leaves a copy in the save area (NfcScanArgs::saveRings) */
   uint32_t saved = 0;

   if (mode != NFC_WAVE_WINDOWS)
   {
      nfc_wave_rings_out(lds, laneRings, pitch, u.s.clock, cc.corrTotal);

      for (uint32_t i = lane; i < NFC_STREAM_BYTES / 4u; i += NFC_LANES)
         ((uint32_t *)(L.bytes + (uint64_t)it.w * NFC_STREAM_BYTES))[i] = ((const NFC_WAVE_LDS uint32_t *)lds->bytes)[i];
   }
   else if (how == 0u && !closing)
   {
      uint32_t slot = 0;

      NFC_WAVE_UNIFORM_BEGIN(u)
      {
         u.at = NFC_ATOMIC_ADD(A.saveNext, 1u);
      }
      NFC_WAVE_UNIFORM_END(u)

      slot = u.at;

      if (slot < A.saveRoom)
      {
         const uint32_t rows = L.ringBlockFloats / NFC_LANES;

         nfc_wave_rings_out(lds, A.saveRings + (uint64_t)slot * rows, 1u, u.s.clock, cc.corrTotal);

         for (uint32_t i = lane; i < NFC_STREAM_BYTES / 4u; i += NFC_LANES)
            ((uint32_t *)(A.saveBytes + (uint64_t)slot * NFC_STREAM_BYTES))[i] = ((const NFC_WAVE_LDS uint32_t *)lds->bytes)[i];

         saved = slot + 1u;
      }
   }

   if (lane == 0)
   {
      L.states[it.w] = u.s;
      me->stop = it.startPos + u.consumed;
      me->retired = how;
      if (mode == NFC_WAVE_WINDOWS)
         me->saved = saved;

      const uint32_t tilesStepped = (u.stepped + NFC_LANES - 1u) / NFC_LANES;
      NFC_WAVE_STAT_ADD(L.laneStats, tilesStepped);
      NFC_WAVE_STAT_MAX(L.laneStats + 1, tilesStepped);
      NFC_WAVE_STAT_ADD(L.laneStats + 2, 1u);
   }
}

#endif
