/*
 * nfc_scan.hpp — device code of the time-parallel path (see nfc_scan.h for the idea and the records).
 *
 *   nfc_scan_*      the chunk walker: exact front end (the very nfc_front_end_core the decoder runs) plus the per-tile
 *                   "can a detector at rest move here?" tests; one lane per chunk.
 *   nfc_seams_*     verifies every chunk seam bit for bit and resolves the edge-tracker time across chunks.
 *   nfc_windows_*   turns tile flags into windows (clusters of busy tiles) and marks where a lane may retire.
 *   nfc_window_*    prepares the lane of a window: scanned front-end state + assumed carry, everything else at rest.
 *   nfc_chain_*     after a decode pass: which lanes' frames are the stream's frames, were their assumptions right.
 *   nfc_finish_*    copies the live frames, in stream order, to the frame sink and the last lane's state to the stream.
 *
 * Included by nfc_kernels.hip after nfc_core.hpp (and by the CPU twins of tests/hostsim, test infrastructure).
 * Reference behaviour: NfcTech.cpp:28-105 (front end), NfcDecoder.cpp:394-418,472-523 (search loop, carrier detector),
 * detectModulation of NfcA.cpp:217-411, NfcB.cpp:238-432, NfcF.cpp:206-408, NfcV.cpp:236-435 (what moves a detector
 * at rest).
 */
#ifndef NFC_AMD_SCAN_HPP
#define NFC_AMD_SCAN_HPP

#include "nfc_scan.h"

/* constants of a scan launch, derived on the host from the configuration of the streams it covers */
struct NfcScanParams
{
   float rangeK;   /* 0.49 * smallest correlation threshold of the enabled raw-signal detectors (A, F, V); +inf when none */
   float edgeK;    /* 0.99 * NFC-B minimum modulation depth; +inf when NFC-B is disabled */
   float deepK;    /* 0.98 * smallest maximum modulation depth of the enabled detectors */
   uint32_t chunkSamples;  /* samples per chunk (multiple of NFC_SCAN_POINT) */
   uint32_t warmSamples;   /* samples walked before a chunk to reach the true front-end state (multiple of NFC_SCAN_POINT) */
   uint32_t soloSamples;   /* streams of at most this many samples get no speculative windows: their carry lane decodes them alone, in one pass */
   uint32_t offGridAlone;  /* a stream with samples off the capture grid is decoded by its carry lane alone (the wave decoder walks the
                              running sums in the step's order there: nfc_wave_fast.hpp); 0: it takes the sequential kernels */
};

/* bit-for-bit equality of two records (word by word through memcpy: no library call on the device, and no loads through
 * an integer alias of float fields, which type-based alias analysis may move across the stores that produced them) */
NFC_DEV bool nfc_same_words(const void *a, const void *b, uint32_t bytes)
{
   const unsigned char *p = (const unsigned char *)a, *q = (const unsigned char *)b;
   bool same = true;

   for (uint32_t i = 0; i + 4 <= bytes; i += 4)
   {
      uint32_t u, v;
      __builtin_memcpy(&u, p + i, 4);
      __builtin_memcpy(&v, q + i, 4);
      same = same && u == v;
   }

   return same;
}

/* walker state of one chunk */
struct NfcScanLane
{
   NfcStreamState fe; /* only the front-end fields are used */
   uint32_t zone;
   uint32_t edgeSynced; /* the edge-peak tracker has been reset since the walk began: its peak is the true one */
   uint32_t edgeKnown;  /* ... and it has set a time since: fe.edgeTime is the true one */
   float xmin, xmax, envmin, envmax, fmin; /* of the tile being walked */
   uint32_t bits;
   float seedMax; /* largest raw sample of the whole tiles walked so far (nfc_scan_reseed) */
};

#define NFC_SCAN_BIG 3.0e38f
#define NFC_ZONE_MASK 0xFFu
#define NFC_ZONE_EDGE_KNOWN 0x100u
#define NFC_ZONE_EDGE_SYNCED 0x200u

NFC_DEV void nfc_scan_tile_reset(NfcScanLane &w)
{
   w.xmin = NFC_SCAN_BIG;
   w.xmax = -NFC_SCAN_BIG;
   w.envmin = NFC_SCAN_BIG;
   w.envmax = -NFC_SCAN_BIG;
   w.fmin = NFC_SCAN_BIG;
   /* the detectors wait for the decoder's first 1024 samples: the tile's first sample gets clock + 1 */
   w.bits = (w.fe.clock + 1u < 1024u) ? NFC_TILE_UNARMED : 0u;
}

/* start of a walk: `exact` = from the stream's own state (first chunk of a submission), else from a guess that the
 * warm-up turns into the true state */
NFC_DEV void nfc_scan_begin(NfcScanLane &w, const NfcStreamState *from, uint32_t clock, float first)
{
   __builtin_memset(&w.fe, 0, sizeof(w.fe));

   if (from)
   {
      w.fe.clock = from->clock;
      w.fe.pulseFilter = from->pulseFilter;
      w.fe.env = from->env;
      w.fe.n1 = from->n1;
      w.fe.mdev = from->mdev;
      w.fe.avg = from->avg;
      w.fe.edgePeak = from->edgePeak;
      w.fe.edgeTime = from->edgeTime;
      w.zone = from->carrierOn ? 1u : (from->carrierOff ? 2u : 0u);
      w.edgeSynced = 1;
      w.edgeKnown = 0; /* the decoder's copy may have been zeroed by a carrier frame: resolved through NfcCarry */
   }
   else
   {
      w.fe.clock = clock;
      w.fe.env = first;
      w.fe.n1 = first * 10.0f; /* fixed point of n = x + 0.9 n */
      w.fe.avg = first;
      w.zone = 0;
      w.edgeSynced = 0;
      w.edgeKnown = 0;
   }

   nfc_scan_tile_reset(w);
}

NFC_DEV void nfc_scan_point(const NfcScanLane &w, NfcScanPoint &p)
{
   p.env = w.fe.env;
   p.n1 = w.fe.n1;
   p.mdev = w.fe.mdev;
   p.avg = w.fe.avg;
   p.edgePeak = w.fe.edgePeak;
   p.pulseFilter = w.fe.pulseFilter;
   p.edgeTime = w.fe.edgeTime;
   p.zone = w.zone | (w.edgeKnown ? NFC_ZONE_EDGE_KNOWN : 0u) | (w.edgeSynced ? NFC_ZONE_EDGE_SYNCED : 0u);
}

/* A third of the way into the warm-up the guessed envelope is replaced by the largest sample seen so far. The tracker
 * only follows the signal inside a 5 % capture range; outside it moves once per ten symbols and can take a whole chunk to
 * find the level. In a modulated signal (pauses of a reader, a card's load modulation) the true tracker sits at the
 * unmodulated carrier level and holds through the modulation: that level is the largest the signal reaches, up to the
 * noise. (The average, used here before, lies between the levels of a modulated signal: started there the tracker settled
 * on whichever level came first, and most seams of a busy capture did not verify.) */
NFC_DEV void nfc_scan_reseed(NfcScanLane &w)
{
   w.fe.env = w.seedMax > 0.0f ? w.seedMax : w.fe.avg;
   w.fe.pulseFilter = 0;
}

/* one sample: the decoder's own front end, then what the tile tests need; returns the DC-removed sample */
NFC_DEV float nfc_scan_sample(const NfcConfig &c, NfcScanLane &w, float x)
{
   ++w.fe.clock;
   ++w.fe.pulseFilter;

   const uint32_t timeBefore = w.fe.edgeTime;

   const NfcNow now = nfc_front_end_core(c, w.fe, x);

   /* edge tracker bookkeeping: a reset (signal below the low threshold) makes the peak the true one whatever the walk
    * started from; a time set after that is the true time */
   w.edgeSynced |= nfc_abs(now.filt) < c.lowThreshold ? 1u : 0u;
   w.edgeKnown |= (w.fe.edgeTime != timeBefore) ? w.edgeSynced : 0u;

   w.xmin = x < w.xmin ? x : w.xmin;
   w.xmax = x > w.xmax ? x : w.xmax;
   w.envmin = w.fe.env < w.envmin ? w.fe.env : w.envmin;
   w.envmax = w.fe.env > w.envmax ? w.fe.env : w.envmax;
   w.fmin = now.filt < w.fmin ? now.filt : w.fmin;

   /* on the int16 grid of the captures (k / 32768; |k| small enough for every box sum to stay exact: nfc_scan_tile_end)?
    * A NaN is not equal to its floor either */
   const float scaled = x * 32768.0f;
   w.bits |= (scaled != __builtin_floorf(scaled)) ? NFC_TILE_OFFGRID : 0u;

   /* NaNs make both comparisons false, exactly as in nfc_detect_carrier */
   {
      const bool up = w.fe.avg > c.highThreshold;
      const bool down = !up && w.fe.avg < c.lowThreshold;
      const uint32_t zone = up ? 1u : (down ? 2u : w.zone);

      w.bits |= zone != w.zone ? NFC_TILE_CARRIER : 0u;
      w.zone = zone;
   }

   return now.filt;
}

/* end of a tile */
NFC_DEV void nfc_scan_tile_end(NfcScanLane &w, NfcScanTile &out)
{
   out.xmin = w.xmin;
   out.xmax = w.xmax;
   out.fmin = w.fmin;
   out.envmin = w.envmin;
   out.envmax = w.envmax;
   out.bits = w.bits | ((w.xmin >= -4.0f && w.xmax <= 4.0f) ? 0u : NFC_TILE_OFFGRID); /* also an empty tile, harmlessly */
   w.seedMax = w.xmax > w.seedMax ? w.xmax : w.seedMax;
   nfc_scan_tile_reset(w);
}

/* The tile tests (nfc_scan.h): flag word of tile i of a job from the recorded extremes of the tile and of the tiles a
 * detector can still look back into. Written so that NaNs (and an envelope of zero or less) come out busy. */
NFC_DEV uint32_t nfc_tile_flags(const NfcConfig &c, const NfcScanParams &sp, const NfcScanTile *t, uint32_t i)
{
   uint32_t flags = t[i].bits;

   if (i < NFC_SCAN_LOOKBACK)
      flags |= NFC_TILE_NOHISTORY;
   else
   {
      float lo = t[i].xmin, hi = t[i].xmax;

      for (uint32_t k = 1; k <= NFC_SCAN_LOOKBACK; k++)
      {
         lo = t[i - k].xmin < lo ? t[i - k].xmin : lo;
         hi = t[i - k].xmax > hi ? t[i - k].xmax : hi;
      }

      if (!((hi - lo) <= sp.rangeK * t[i].envmin))
         flags |= NFC_TILE_RANGE;

      /* modulation deeper than the smallest maximum a detector accepts (NfcF.cpp:262: the one test of a search detector
       * that is not a correlation; it clears a record that is not at rest): depth = (env - x) / env <= (envmax - xmin) / envmin */
      {
         float xm = t[i].xmin < t[i - 1].xmin ? t[i].xmin : t[i - 1].xmin;
         float eh = t[i].envmax > t[i - 1].envmax ? t[i].envmax : t[i - 1].envmax;
         float el = t[i].envmin < t[i - 1].envmin ? t[i].envmin : t[i - 1].envmin;
         if (!((eh - xm) <= sp.deepK * el))
            flags |= NFC_TILE_RANGE;
      }

      float fm = t[i].fmin;
      for (uint32_t k = 1; k <= NFC_SCAN_EDGEBACK; k++)
         fm = t[i - k].fmin < fm ? t[i - k].fmin : fm;

      if (!(fm >= -(sp.edgeK * t[i].envmin)))
         flags |= NFC_TILE_EDGE;
   }

   if (!(t[i].envmin >= c.powerThreshold))
      flags |= NFC_TILE_UNARMED;

   /* dark: the detector bank is not stepped on any sample of the tile and the carrier detector has nothing to do */
   if (t[i].envmax < c.powerThreshold && !(t[i].bits & (NFC_TILE_CARRIER | NFC_TILE_OFFGRID)))
      flags |= NFC_TILE_DARK;

   return flags;
}

/* ------------------------------------------------------------------------------------------ */
/* seams: one thread per job, chunks in order                                                  */
/* ------------------------------------------------------------------------------------------ */

/* The includer says how to read sample i of a stream as a magnitude: NFC_SAMPLE_AT(data, stride, i). */
#ifndef NFC_SAMPLE_AT
#error "define NFC_SAMPLE_AT(data, stride, index) before including nfc_scan.hpp"
#endif

/* the recurrences of a point, bit for bit (not the edge time, which a walk may not know: see NFC_ZONE_EDGE_KNOWN) */
NFC_DEV bool nfc_point_same(const NfcScanPoint &a, const NfcScanPoint &b)
{
   /* bit patterns: env n1 mdev avg edgePeak pulseFilter are the first six words of a point */
   return nfc_same_words(&a, &b, 6 * sizeof(uint32_t)) && ((a.zone ^ b.zone) & NFC_ZONE_MASK) == 0;
}

/* continue a walk from a known state */
NFC_DEV void nfc_scan_resume(NfcScanLane &w, const NfcScanPoint &p, uint32_t edgeTime, uint32_t clock)
{
   __builtin_memset(&w.fe, 0, sizeof(w.fe));
   w.fe.clock = clock;
   w.fe.pulseFilter = p.pulseFilter;
   w.fe.env = p.env;
   w.fe.n1 = p.n1;
   w.fe.mdev = p.mdev;
   w.fe.avg = p.avg;
   w.fe.edgePeak = p.edgePeak;
   w.fe.edgeTime = edgeTime;
   w.zone = p.zone & NFC_ZONE_MASK;
   w.edgeSynced = 1;
   w.edgeKnown = 1;
   nfc_scan_tile_reset(w);
}

/* Seams of a job, chunks in order (one thread). A chunk whose walk did not start from the state the chunk before ends
 * with - the envelope tracker is not contractive, the other recurrences sometimes need longer than the warm-up - has to
 * be walked again from that state. Every such chunk of the job goes on the repair list at once, each to start from the
 * end its predecessor has on record now: a second walk follows the first one's trajectory from the sample on at which
 * the two states agree bit for bit (nfc_scan_merged: it stops there and the chunk's end stands), so a predecessor that is
 * itself being walked again usually keeps its end. Where it does not, this check - run again after every round of walks -
 * finds the successor's start unequal to the new end and lists it again. At the fixed point every chunk starts from its
 * predecessor's end and the first from the stream's own state: every record is the true one, whatever the warm-ups and
 * the guesses achieved. Returns true when the job is waiting for repairs.
 * chunkEdge[k] = edge-tracker time at the start of chunk k (points without a time of their own inherit it). */
NFC_DEV bool nfc_seams_check(NfcScanJob &job, uint32_t jobIndex, NfcScanSeam *seams, uint32_t *chunkEdge, uint32_t startEdge, NfcScanChunk *repairs,
                             uint32_t *repairCount, NfcScanPoint *points = nullptr, uint32_t chunkSamples = 0, NfcScanChunk *repairsEnv = nullptr,
                             uint32_t *repairEnvCount = nullptr, uint32_t *stale = nullptr)
{
   uint32_t edge = startEdge; /* edge time at the start of the chunk at hand, by the records as they are */
   bool pending = false;
   bool chain = false; /* the chunk before is listed this round, its envelope tracker alone to be walked again */

   for (uint32_t k = 0; k < job.chunks; k++)
   {
      NfcScanSeam &s = seams[job.firstChunk + k];
      bool listedEnvelope = false;

      const bool sound = k == 0 || (nfc_point_same(s.start, seams[job.firstChunk + k - 1].end) &&
                                    (!(s.start.zone & NFC_ZONE_EDGE_KNOWN) || s.start.edgeTime == edge));

      chunkEdge[job.firstChunk + k] = edge;

#ifdef NFC_SEAM_DEBUG
      if (!sound)
         NFC_SEAM_DEBUG(k, s.start, seams[job.firstChunk + k - 1].end, edge);
#endif

      if (!sound && points)
      {
         /* The carrier zone is a latch: the zone the average was last seen in, kept while the average lies between the two
          * thresholds. A walk that has not seen the average outside them yet (zone 0: its warm-up and the chunk up to some
          * sample lie in between - a weak capture can stay there for a million samples) differs from the true walk in
          * nothing but that: where it says "none yet" the true zone is the one the chunk before ended in. That is put
          * right here, without a walk: the chunk's start, its stored points up to the first that has a zone, its end if it
          * never found one. (The tile flag of the first sample outside the thresholds may then mark a change of zone that
          * is none: a tile looked at for nothing.) Within this loop the corrected end is what the next chunk is held
          * against, so a run of such chunks settles in one check - they used to be walked again one per round, every
          * field of them. */
         const NfcScanPoint &before = seams[job.firstChunk + k - 1].end;
         const uint32_t zoneBefore = before.zone & NFC_ZONE_MASK;

         if ((s.start.zone & NFC_ZONE_MASK) == 0u && zoneBefore != 0u && nfc_bits(s.start.n1) == nfc_bits(before.n1) && nfc_bits(s.start.mdev) == nfc_bits(before.mdev) &&
             nfc_bits(s.start.avg) == nfc_bits(before.avg) && nfc_bits(s.start.edgePeak) == nfc_bits(before.edgePeak))
         {
            const uint32_t first = k * chunkSamples;
            const uint32_t last = first + chunkSamples < job.count ? first + chunkSamples : job.count;

            s.start.zone |= zoneBefore;

            bool open = true; /* no zone of the chunk's own yet */

            for (uint32_t pos = first; pos < last && open; pos += NFC_SCAN_POINT)
            {
               NfcScanPoint &p = points[job.firstPoint + pos / NFC_SCAN_POINT];

               if ((p.zone & NFC_ZONE_MASK) == 0u)
                  p.zone |= zoneBefore;
               else
                  open = false;
            }

            if ((s.end.zone & NFC_ZONE_MASK) == 0u)
               s.end.zone |= zoneBefore;
         }
      }

      const bool soundNow = sound || (nfc_point_same(s.start, seams[job.firstChunk + k - 1].end) &&
                                      (!(s.start.zone & NFC_ZONE_EDGE_KNOWN) || s.start.edgeTime == edge));

      if (!soundNow)
      {
         const NfcScanPoint &before = seams[job.firstChunk + k - 1].end;

         /* Usually it is the envelope tracker alone that started wrong (it is the one recurrence that is not contractive:
          * after a level step it sits between the levels for tens of thousands of samples, a walk seeded from the signal
          * does not). Nothing else in the front end reads the envelope: the rest of the chunk's records stand, and the
          * second walk is the tracker's alone (a tenth of the instructions). */
         const bool envelopeOnly = nfc_bits(s.start.n1) == nfc_bits(before.n1) && nfc_bits(s.start.mdev) == nfc_bits(before.mdev) &&
                                   nfc_bits(s.start.avg) == nfc_bits(before.avg) && nfc_bits(s.start.edgePeak) == nfc_bits(before.edgePeak) &&
                                   ((s.start.zone ^ before.zone) & NFC_ZONE_MASK) == 0 && (!(s.start.zone & NFC_ZONE_EDGE_KNOWN) || s.start.edgeTime == edge);

         if (envelopeOnly)
         {
            s.start.env = before.env;
            s.start.pulseFilter = before.pulseFilter;
            listedEnvelope = true;
         }
         else
         {
            /* from its predecessor's end, edge time included */
            s.start = before;
            s.start.edgeTime = edge;
            s.start.zone |= NFC_ZONE_EDGE_KNOWN | NFC_ZONE_EDGE_SYNCED;
         }

         /* (a list of their own for the chunks whose envelope tracker alone is walked again, when the caller keeps one:
          * nfc_envelope_kernel takes those) */
         NfcScanChunk &r = (envelopeOnly && repairsEnv) ? repairsEnv[NFC_ATOMIC_ADD(repairEnvCount, 1u)] : repairs[NFC_ATOMIC_ADD(repairCount, 1u)];
         r.job = jobIndex;
         r.index = k | NFC_CHUNK_REPAIR | (envelopeOnly ? NFC_CHUNK_ENVELOPE : 0u);

         /* (its start state has just changed: planes written from the old one are written again, NfcScanArgs::planesStale) */
         if (stale)
            stale[job.firstChunk + k] = 1u;

         pending = true;
      }

      /* Chunks inherit a wrong envelope from each other in chains, and a chain used to be walked a chunk per round: the walk of
       * chunk k ends on another envelope than the one on record, which the NEXT check finds chunk k + 1 not to start from. The
       * marks let nfc_envelope_kernel settle a chain in one walk: a walk that ends a chunk on another envelope than the next
       * chunk starts from goes on into that chunk - unless somebody else walks it this round (NFC_ZONE_LISTED: on a list, and
       * not as the follower of the chunk before it, NFC_ZONE_FOLLOWS) - until it meets the trajectory on record. (The scan
       * kernel, when it is given the list, walks every listed chunk on its own as before.) */
      if (listedEnvelope && chain)
         s.start.zone |= NFC_ZONE_FOLLOWS;
      else
         s.start.zone &= ~NFC_ZONE_FOLLOWS;

      if (!soundNow)
         s.start.zone |= NFC_ZONE_LISTED;
      else
         s.start.zone &= ~NFC_ZONE_LISTED;

      chain = listedEnvelope;

      /* edge time at the end of the chunk (a chunk about to be walked again may change it: the next round looks again) */
      if (s.end.zone & NFC_ZONE_EDGE_KNOWN)
         edge = s.end.edgeTime;
   }

   return pending;
}

/* A second walk has reached the stored point `was` of its chunk in state `now`: from here on it would repeat the first
 * walk sample for sample (the front end depends on nothing else). */
NFC_DEV bool nfc_scan_merged(const NfcScanPoint &now, const NfcScanPoint &was)
{
   return nfc_point_same(now, was);
}

/* What the first walk recorded after the merge stands, except that it may not have trusted its edge time: the tracker's
 * time at a later point is the first walk's if that has moved since the merge (an update made in the merged state is the
 * true one), else the time the second walk brings (`edge`). `atMerge`: the first walk's time at the merge. */
NFC_DEV void nfc_scan_adopt(NfcScanPoint &p, uint32_t atMerge, uint32_t edge)
{
   p.edgeTime = p.edgeTime != atMerge ? p.edgeTime : edge;
   p.zone |= NFC_ZONE_EDGE_KNOWN | NFC_ZONE_EDGE_SYNCED;
}

/* ------------------------------------------------------------------------------------------ */
/* windows: one thread per job                                                                 */
/* ------------------------------------------------------------------------------------------ */

/* Marks the tiles from which on at least NFC_WINDOW_GAP tiles are quiet (a lane may retire there: whatever comes next
 * is far enough for the following lane's warm-up to lie in quiet signal) and lists the speculative windows of the job:
 * one per busy tile that follows such a gap, plus a closing one that activates at the end of the submission (it decodes
 * nothing: it rebuilds the rings a stream at rest has there, for the case that the last lane retired before the end).
 * The end of the submission does not count as a gap: a lane still running in the last quiet stretch runs to the end.
 * `room` windows may be written at `out`; returns the number the job needs. */
NFC_DEV uint32_t nfc_windows_build(const NfcScanJob &job, uint32_t jobIndex, uint32_t *flags, NfcWindow *out, uint32_t room)
{
   const uint32_t nTiles = (job.count + NFC_SCAN_TILE - 1) / NFC_SCAN_TILE;
   uint32_t *t = flags + job.firstTile;

   uint32_t quietAhead = 0;
   uint32_t darkAhead = 0;

   for (uint32_t i = nTiles; i-- > 0;)
   {
      darkAhead = (t[i] & NFC_TILE_DARK) ? (darkAhead < 0xFFFFu ? darkAhead + 1u : darkAhead) : 0u;
      t[i] = (t[i] & 0xFFFFu) | (darkAhead << NFC_TILE_DARK_RUN_SHIFT);

      if (t[i] & NFC_TILE_BUSY)
         quietAhead = 0;
      else if (quietAhead < NFC_WINDOW_GAP)
         quietAhead++;

      if (quietAhead >= NFC_WINDOW_GAP)
         t[i] |= NFC_TILE_RETIRE_OK;
      else
         t[i] &= ~NFC_TILE_RETIRE_OK;
   }

   uint32_t n = 0;

   auto put = [&](uint32_t start, uint32_t activate) {
      if (n < room)
      {
         NfcWindow &w = out[n];
         __builtin_memset(&w, 0, sizeof(w));
         w.job = jobIndex;
         w.start = start;
         w.activate = activate;
         w.verify = activate + NFC_WINDOW_VERIFY + NFC_SCAN_TILE <= job.count ? activate + NFC_WINDOW_VERIFY : 0xFFFFFFFFu;
      }
      n++;
   };

   const uint32_t warm = NFC_WINDOW_WARM_FRONT + NFC_WINDOW_WARM_CORR;

   uint32_t quietBehind = 0;
   uint32_t lastAct = 0; /* activation of the most recent window (the carry lane goes live at 0) */

   for (uint32_t i = 0; i < nTiles; i++)
   {
      const uint32_t act = i * NFC_SCAN_TILE;

      if (!(t[i] & NFC_TILE_BUSY))
         quietBehind++;
      else
      {
         /* a busy tile after a gap a lane may have retired in */
         if (quietBehind >= NFC_WINDOW_GAP && act >= warm + NFC_SCAN_POINT)
         {
            put((act - warm) / NFC_SCAN_POINT * NFC_SCAN_POINT, act);
            lastAct = act;
         }

         quietBehind = 0;
      }

      /* where lanes cannot retire (busy signal, or not enough quiet signal ahead) a new lane starts every
       * NfcScanJob::cut samples: whoever is running there hands over to it if their states agree (nfc_lane_handover) */
      if (!(t[i] & (NFC_TILE_RETIRE_OK | NFC_TILE_DARK)) && act >= lastAct + job.cut && act >= warm + NFC_SCAN_POINT &&
          act + NFC_WINDOW_VERIFY + NFC_SCAN_TILE <= job.count)
      {
         put((act - warm) / NFC_SCAN_POINT * NFC_SCAN_POINT, act);
         lastAct = act;
      }
   }

   if (job.count >= warm + NFC_SCAN_POINT && job.count > lastAct)
      put((job.count - warm) / NFC_SCAN_POINT * NFC_SCAN_POINT, job.count);

   return n;
}

/* ---- the same, 64 tiles at a time ----
 * nfc_windows_build is the statement of the rule; one thread walking a million tiles of a long capture took longer than
 * decoding it. The kernel (one wave per job) forms bit masks of 64 tiles with ballots and applies the rule to the masks:
 * bit l of a mask = tile 64 * g + l. The emulated runtime runs both forms and compares them. */

/* bits l .. l+63 of the 128-bit string hi:lo */
NFC_DEV uint64_t nfc_bits_from(uint64_t lo, uint64_t hi, uint32_t l)
{
   return l == 0u ? lo : ((lo >> l) | (hi << (64u - l)));
}

/* a lane may retire at tile l of the group: tiles l .. l+15 exist and are not busy (`blocked`: busy or beyond the end) */
NFC_DEV bool nfc_group_retire_ok(uint64_t blockedHere, uint64_t blockedNext, uint32_t l)
{
   return (nfc_bits_from(blockedHere, blockedNext, l) & ((1ull << NFC_WINDOW_GAP) - 1ull)) == 0ull;
}

/* dark tiles in a row from tile l of the group on; `carry`: the same for the first tile of the next group (0 after the last) */
NFC_DEV uint32_t nfc_group_dark_run(uint64_t dark, bool full, uint32_t l, uint32_t carry)
{
   const uint64_t inv = ~(dark >> l);
   uint32_t run = inv ? (uint32_t)__builtin_ctzll(inv) : 64u;
   if (run > 64u - l)
      run = 64u - l;
   if (run == 64u - l && full)
      run += carry;
   return run > 0xFFFFu ? 0xFFFFu : run;
}

/* a window starts at busy tile l of the group when the 16 tiles before it exist and are quiet
 * (`busyBefore`: the group before; all ones before the first group) */
NFC_DEV bool nfc_group_cluster(uint64_t busyBefore, uint64_t busyHere, uint32_t l)
{
   if (!((busyHere >> l) & 1ull))
      return false;
   const uint64_t w = l >= NFC_WINDOW_GAP ? (busyHere >> (l - NFC_WINDOW_GAP)) : ((busyBefore >> (64u - NFC_WINDOW_GAP + l)) | (busyHere << (NFC_WINDOW_GAP - l)));
   return (w & ((1ull << NFC_WINDOW_GAP) - 1ull)) == 0ull;
}

/* static conditions of a window that activates at tile i */
NFC_DEV bool nfc_tile_may_start(uint32_t i, uint32_t count)
{
   return i * NFC_SCAN_TILE >= NFC_WINDOW_WARM_FRONT + NFC_WINDOW_WARM_CORR + NFC_SCAN_POINT;
}

NFC_DEV bool nfc_tile_may_cut(uint32_t i, uint32_t count)
{
   return nfc_tile_may_start(i, count) && i * NFC_SCAN_TILE + NFC_WINDOW_VERIFY + NFC_SCAN_TILE <= count;
}

/* a window record as the builder leaves it: everything zero but these four words. (The includer may have the lanes of a wave
 * share the stores - nfc_kernels.hip: the record is 1.4 KB, and one lane cleared it a word at a time, 180 000 records a step of
 * the headline.) */
#ifndef NFC_WINDOW_STORE
#define NFC_WINDOW_STORE(w, job_, start_, activate_, verify_) \
   do                                                        \
   {                                                         \
      __builtin_memset(&(w), 0, sizeof(w));                  \
      (w).job = (job_);                                      \
      (w).start = (start_);                                  \
      (w).activate = (activate_);                            \
      (w).verify = (verify_);                                \
   } while (0)
#endif

struct NfcWindowPlacer
{
   uint32_t lastAct; /* activation of the most recent window (the carry lane goes live at 0) */
   uint32_t n;       /* windows so far */
};

NFC_DEV void nfc_window_put(NfcWindowPlacer &p, const NfcScanJob &job, uint32_t jobIndex, NfcWindow *out, uint32_t room, uint32_t start, uint32_t activate, bool write)
{
   if (write && p.n < room)
   {
      const uint32_t verify = activate + NFC_WINDOW_VERIFY + NFC_SCAN_TILE <= job.count ? activate + NFC_WINDOW_VERIFY : 0xFFFFFFFFu;
      NFC_WINDOW_STORE(out[p.n], jobIndex, start, activate, verify);
   }
   p.n++;
}

/* windows of group g: `cluster` = tiles where a window starts after a gap (with nfc_tile_may_start), `cut` = tiles where a
 * lane cannot retire (with nfc_tile_may_cut). Same order of decisions as the loop of nfc_windows_build. */
NFC_DEV void nfc_group_place(NfcWindowPlacer &p, const NfcScanJob &job, uint32_t jobIndex, NfcWindow *out, uint32_t room, uint32_t g, uint64_t cluster,
                             uint64_t cut, bool write)
{
   const uint32_t warm = NFC_WINDOW_WARM_FRONT + NFC_WINDOW_WARM_CORR;
   uint32_t pos = 0;

   while (pos < 64u)
   {
      const uint64_t from = ~0ull << pos;
      const uint64_t c = cluster & from;

      /* first tile of the group a cut window may activate at */
      const uint32_t due = (p.lastAct + job.cut) / NFC_SCAN_TILE;
      uint64_t e = 0ull;
      if (due < g * 64u + 64u)
      {
         const uint32_t first = due > g * 64u + pos ? due - g * 64u : pos;
         e = cut & (~0ull << first);
      }

      if (!c && !e)
         break;

      const uint32_t lc = c ? (uint32_t)__builtin_ctzll(c) : 64u;
      const uint32_t le = e ? (uint32_t)__builtin_ctzll(e) : 64u;
      const uint32_t l = lc <= le ? lc : le;
      const uint32_t act = (g * 64u + l) * NFC_SCAN_TILE;

      nfc_window_put(p, job, jobIndex, out, room, (act - warm) / NFC_SCAN_POINT * NFC_SCAN_POINT, act, write);
      p.lastAct = act;
      pos = l + 1u;
   }
}

/* the closing window */
NFC_DEV void nfc_windows_close(NfcWindowPlacer &p, const NfcScanJob &job, uint32_t jobIndex, NfcWindow *out, uint32_t room, bool write)
{
   const uint32_t warm = NFC_WINDOW_WARM_FRONT + NFC_WINDOW_WARM_CORR;

   if (job.count >= warm + NFC_SCAN_POINT && job.count > p.lastAct)
      nfc_window_put(p, job, jobIndex, out, room, (job.count - warm) / NFC_SCAN_POINT * NFC_SCAN_POINT, job.count, write);
}

/* ------------------------------------------------------------------------------------------ */
/* carry: what a lane inherits / leaves                                                        */
/* ------------------------------------------------------------------------------------------ */

NFC_DEV void nfc_carry_take(NfcCarry &x, const NfcStreamState &s, const NfcStreamCold &cold)
{
   for (int t = 0; t < 4; t++)
      x.tim[t] = cold.tim[t];
   x.chainedA = s.chainedA;
   x.carrierOn = s.carrierOn;
   x.carrierOff = s.carrierOff;
   x.emitClock = cold.emitClock;
   x.emitValid = cold.emitValid;
   x.edgeTime = s.edgeTime;
   x.waitOwn = cold.waitFlags;
   x.emitOwn = cold.emitOwn;

   /* only meaningful (and only looked at) for a lane that stopped at rest: the detector records share their storage
    * with the decode registers */
   for (int i = 0; i < 2; i++)
   {
      x.pulsesF[i] = s.lockTech ? 0u : s.u.search.detF[i].pulses;
      x.thrF[i] = s.lockTech ? 0.0f : s.u.search.detF[i].thr;
      x.clearedF[i] = cold.clearedF[i];
      x.ownF[i] = (cold.boundF[i].flags & NFC_FBOUND_THR_OWN) ? 1u : 0u;
   }

   if (s.lockTech)
      __builtin_memset(&x.search, 0, sizeof(x.search));
   else
      x.search = s.u.search;
}

/* detector records as far as they decide anything: without the running sums (their offset never shows), without the
 * two NFC-F leftovers that are compared on their own (pulsesF / thrF), and without fields an idle detector rewrites
 * before it reads them (nfc_at_rest) */
NFC_DEV void nfc_records_canonical(NfcSearchRegs &r)
{
   for (int i = 0; i < 3; i++)
      r.detA[i].acc = 0.0f;

   for (int i = 0; i < 2; i++)
   {
      NfcDetB &b = r.detB[i];
      if ((b.symStart | b.symEnd | b.winStart | b.winEnd | b.auxTime | nfc_bits(b.aux)) == 0)
         b.thr = 0.0f;
   }

   for (int i = 0; i < 2; i++)
   {
      NfcDetF &f = r.detF[i];
      f.acc = 0.0f;
      f.pulses = 0;
      f.thr = 0.0f;
      if ((f.winStart | f.winEnd | f.sync | f.symStart | f.symEnd | f.peakTime | nfc_bits(f.peak)) == 0)
      {
         f.lastPhase = 0.0f;
         f.lastValue = 0.0f;
         f.syncValue = 0.0f;
         f.c0 = 0.0f;
      }
   }

   r.detV.acc = 0.0f;
}

NFC_DEV bool nfc_records_same(const NfcSearchRegs &a, const NfcSearchRegs &b, uint32_t used = 0xFFFFFu)
{
   NfcSearchRegs x = a, y = b;
   nfc_records_canonical(x);
   nfc_records_canonical(y);

   /* an NFC-F record the lane never ran its tracker on before clearing it need not agree (NfcStreamCold::usedTech) */
   for (int i = 0; i < 2; i++)
   {
      if (!((used >> (14 + i)) & 1u))
         y.detF[i] = x.detF[i];
   }

   /* memcmp, not words through an integer alias of the float fields (type-based alias analysis would let the loads
    * pass the stores of the canonical form) */
   return nfc_same_words(&x, &y, sizeof(NfcSearchRegs));
}

/* The fields a later decode can depend on. guardTime / waitingTime are rewritten by every poll frame's processing before
 * they are read (nfc*_process), the carrier times only matter as "set or not" (NfcDecoder.cpp:449-463,472-523). */
/* the decoder's edge time at a lane's first sample: the tracker's, unless a carrier frame has zeroed it since */
NFC_DEV uint32_t nfc_edge_time(const NfcCarry &x, uint32_t tracked)
{
   return (x.emitValid && (int32_t)(tracked - x.emitClock) <= 0) ? 0u : tracked;
}

/* The pulse memory of an NFC-F preamble detector (counter, threshold of the last pulse) a lane ran on - `had`: what it
 * assumed at its start, or what it held at the sample it took over at - against what the stream really held there (`have`).
 * The memory is read at the evaluation of a pulse only (nfcf_track_preamble), and there only through two comparisons: the
 * counter against 94 and the pulse against the threshold; the counter is counted on from whatever it was, the threshold is
 * replaced by the first pulse that is accepted or clears the record. The lane has recorded what its evaluations found
 * (NfcFBound, until the record started over): any other memory that puts every one of those comparisons on the same side
 * would have made the lane decide, emit and leave exactly the same, up to the counter's offset and a threshold it never
 * replaced - which nfc_chain_follow puts right in what the lane leaves. `startedOver` / `ownThreshold`: at a meeting
 * sample, whether the lane's counter / threshold were no longer the assumed ones there (then they have to be the same). */
NFC_DEV bool nfc_fbound_admits(const NfcFBound &b, uint32_t hadPulses, float hadThr, uint32_t havePulses, float haveThr, bool startedOver, bool ownThreshold)
{
   const bool samePulses = hadPulses == havePulses;
   const bool sameThr = nfc_bits(hadThr) == nfc_bits(haveThr);

   if (samePulses && sameThr)
      return true;

   if (b.flags & NFC_FBOUND_EXACT)
      return false;

   if (!samePulses)
   {
      if (startedOver)
         return false;

      /* an evaluation that found k found k + d in truth */
      const int64_t d = (int64_t)havePulses - (int64_t)hadPulses;

      if (b.lowMax && !((int64_t)(b.lowMax - 1u) + d < 94))
         return false;
      if (b.highMin && !((int64_t)(b.highMin - 1u) + d >= 94))
         return false;
   }

   if (!sameThr)
   {
      if (ownThreshold)
         return false;
      if ((b.flags & NFC_FBOUND_ABOVE) && !(haveThr > b.thrAbove))
         return false;
      if ((b.flags & NFC_FBOUND_BELOW) && !(haveThr < b.thrBelow))
         return false;
   }

   return true;
}

/* The waiting time of technology t's protocol a lane ran on (`had`: what it assumed at its start, or what it held at the sample
 * it took over at - then `ownThen`: it had set the value itself by that sample) against what the stream really held there
 * (`have`). The value reaches the decode through the comparison `clock > waitingEnd` alone (nfc*_listen_*start), and the lane
 * has noted what its waits on the inherited value required (NfcStreamCold::waitUsed / waitFlags, nfc_wait_ended): none of them
 * ran out, and all of them were over within `waitUsed` of the sample the waiting time counts from, so any value of at least
 * that much leaves every one of those comparisons as it was - the lane would have decided, emitted and left the same, up to
 * the value itself where it never set it (put right in what it hands on: nfc_chain_follow, nfc_final_fixup). `midWait`: the
 * lane's last state is inside a wait on the inherited value (its waitingEnd is part of what it leaves): not covered. */
NFC_DEV bool nfc_wait_admits(const NfcStreamCold &cold, uint32_t t, uint32_t had, uint32_t have, bool ownThen, bool midWait)
{
   if (had == have)
      return true;

   if (ownThen || midWait || ((cold.waitFlags >> (8u + t)) & 1u))
      return false;

   return have >= cold.waitUsed[t];
}

/* Two carries of lanes meeting at the same sample (`meeting`: their decoders' edge times are compared as they are), or
 * an assumption against what the stream holds at a lane's first sample (`tracked`: edge-tracker time there; the last
 * carrier frame only matters through the edge time it leaves the lane with). */
/* `used`: the technologies whose protocol state the lane in question has looked at (NfcStreamCold::usedTech); the others'
 * need not agree. */
NFC_DEV bool nfc_carry_same(const NfcCarry &a, const NfcCarry &b, bool meeting, uint32_t tracked, uint32_t used = 0x3FFFFu)
{
   /* protocol state of a technology the lane never locked, or whose first frame in the lane started the protocol over:
    * nothing the lane did depended on it */
   const uint32_t matters = used & ~(used >> 22) & 0xFu;

   bool same = ((matters & 1u) == 0u || a.chainedA == b.chainedA) && (a.carrierOn != 0) == (b.carrierOn != 0) && (a.carrierOff != 0) == (b.carrierOff != 0) &&
               (meeting ? a.edgeTime == b.edgeTime : nfc_edge_time(a, tracked) == nfc_edge_time(b, tracked));

   for (int t = 0; t < 4; t++)
      same = same && (((matters >> t) & 1u) == 0u ||
                      ((((used >> (4 + t)) & 1u) == 0u || a.tim[t].lastCommand == b.tim[t].lastCommand) && a.tim[t].maxFrameSize == b.tim[t].maxFrameSize &&
                       a.tim[t].protoGuardTime == b.tim[t].protoGuardTime && a.tim[t].protoWaitingTime == b.tim[t].protoWaitingTime));

   for (int i = 0; i < 2; i++)
      same = same && (((used >> (12 + i)) & 1u) == 0u || (a.pulsesF[i] == b.pulsesF[i] && nfc_bits(a.thrF[i]) == nfc_bits(b.thrF[i])));

   same = same && nfc_records_same(a.search, b.search, used);

#ifdef NFC_CARRY_DEBUG
   if (!same)
      NFC_CARRY_DEBUG(a, b, meeting, tracked, used);
#endif

   return same;
}

/* Prediction of what a lane will leave when it is run again with `given` instead of `assumed`, from what it `left`
 * last time: every field it left as it had found it is taken to be passed through, every other to be set by the lane
 * whatever it is given. Only used to choose what to run next; results are always checked. */
NFC_DEV void nfc_carry_predict(NfcCarry &left, const NfcCarry &assumed, const NfcCarry &given, uint32_t used = 0u)
{
#define NFC_CARRY_FIELD(f) left.f = (left.f == assumed.f) ? given.f : left.f
   /* (a lane whose first NFC-A frame was REQA / WUPA has written the chaining flags and every field of the technology's
    * timing itself - `used` bit 22 + t, nfc_finish_frame -: they are its own whatever it is given) */
   if (!((used >> 22) & 1u))
      NFC_CARRY_FIELD(chainedA);

   /* carrier state: only "set or not" is ever read; a lane that emitted no carrier frame passes everything through */
   if (left.emitClock == assumed.emitClock && left.emitValid == assumed.emitValid)
   {
      left.carrierOn = given.carrierOn;
      left.carrierOff = given.carrierOff;
      left.emitClock = given.emitClock;
      left.emitValid = given.emitValid;
   }

   for (int t = 0; t < 4; t++)
   {
      /* (what the lane is known to have written itself - `used` bit 8 + t: the last command; NfcCarry::waitOwn bit 4 + t: the
       * protocol's waiting time - is its own whatever it is given, also where it happens to be what it had assumed) */
      if ((used >> (22 + t)) & 1u)
         continue;
      if (!((used >> (8 + t)) & 1u))
         NFC_CARRY_FIELD(tim[t].lastCommand);
      NFC_CARRY_FIELD(tim[t].guardTime);
      NFC_CARRY_FIELD(tim[t].waitingTime);
      NFC_CARRY_FIELD(tim[t].maxFrameSize);
      NFC_CARRY_FIELD(tim[t].protoGuardTime);
      if (!((left.waitOwn >> (4 + t)) & 1u) || used == 0u)
         NFC_CARRY_FIELD(tim[t].protoWaitingTime);
   }

   for (int i = 0; i < 2; i++)
   {
      /* the pulse counter is counted on: a lane adds what it added before, unless it cleared the counter on the way */
      if (!left.clearedF[i])
         left.pulsesF[i] = given.pulsesF[i] + (left.pulsesF[i] - assumed.pulsesF[i]);
      if (!left.ownF[i] || used == 0u)
         left.thrF[i] = nfc_bits(left.thrF[i]) == nfc_bits(assumed.thrF[i]) ? given.thrF[i] : left.thrF[i];
   }

   /* detector records, one by one: left as found -> whatever it is given */
   {
      NfcSearchRegs l = left.search, a = assumed.search;
      nfc_records_canonical(l);
      nfc_records_canonical(a);

#define NFC_CARRY_RECORD(f)                                                         \
      {                                                                             \
         if (nfc_same_words(&l.f, &a.f, sizeof(l.f)))                               \
            left.search.f = given.search.f;                                         \
      }
      NFC_CARRY_RECORD(detA[0]) NFC_CARRY_RECORD(detA[1]) NFC_CARRY_RECORD(detA[2])
      NFC_CARRY_RECORD(detB[0]) NFC_CARRY_RECORD(detB[1])
      /* (round 5: an NFC-F record the lane has reset on its way - `used`: NfcStreamCold::usedTech bit 16 + i, a mark a reset leaves
       * on a clear record too - is its own whatever it is given. A lane that had found the record at rest and left it at rest was
       * taken to pass on the one with a window in the past the stream really held: the lane after it was told to assume that
       * one, wrongly, and a stream of config 5 went through five passes, a link of the chain each, where three will do.) */
      if (!((used >> 16) & 1u))
         NFC_CARRY_RECORD(detF[0])
      if (!((used >> 17) & 1u))
         NFC_CARRY_RECORD(detF[1])
      NFC_CARRY_RECORD(detV)
#undef NFC_CARRY_RECORD
   }
#undef NFC_CARRY_FIELD
}

/* ------------------------------------------------------------------------------------------ */
/* lane of a window                                                                            */
/* ------------------------------------------------------------------------------------------ */

/* ring phase labels: a lane numbers its correlation rings from zero at its first sample, whatever its clock, so that
 * the lanes of a wave walk the same ring rows (nfc_types.h: NfcStreamCold::label) */
NFC_DEV uint32_t nfc_label(uint32_t clock, uint32_t delay, uint32_t period, uint32_t pos)
{
   /* label = (pos - true position) mod period, true position = (1024 - delay + clock) % period in 32-bit arithmetic */
   const uint32_t truePos = (uint32_t)(1024u - delay + clock) % period;
   return (pos + period - truePos) % period;
}

/* first guess of a lane's carry: what the stream's state holds as the submission finds it, with the carrier state the
 * scan saw at the lane's first sample (a decoder that has been searching knows the carrier is on when the average is) */
NFC_DEV void nfc_carry_guess(NfcCarry &x, const NfcCarry &stream, const NfcScanPoint &p)
{
   x = stream;

   /* detectors at rest (what the stream's state holds belongs to the sample before the submission, not to this lane's) */
   __builtin_memset(&x.search, 0, sizeof(x.search));

   const uint32_t zone = p.zone & NFC_ZONE_MASK;

   if (zone == 1u && !x.carrierOn)
   {
      x.carrierOn = 1u;
      x.carrierOff = 0u;
   }
   else if (zone == 2u && !x.carrierOff)
   {
      x.carrierOff = 1u;
      x.carrierOn = 0u;
   }
}

NFC_DEV void nfc_window_lane(const NfcConfig &c, const NfcWindow &w, const NfcScanPoint &p, uint32_t startClock,
                             NfcStreamState &s, NfcStreamCold &cold)
{
   __builtin_memset(&s, 0, sizeof(s));
   __builtin_memset(&cold, 0, sizeof(cold));

   /* state after the sample before w.start */
   s.clock = startClock + w.start;
   s.pulseFilter = p.pulseFilter;
   s.env = p.env;
   s.n1 = p.n1;
   s.mdev = p.mdev;
   s.avg = p.avg;
   s.edgePeak = p.edgePeak;

   /* the decoder's edge time: the tracker's, unless a carrier frame zeroed it since */
   s.edgeTime = nfc_edge_time(w.carry, w.tracked);

   s.carrierOn = w.carry.carrierOn;
   s.carrierOff = w.carry.carrierOff;
   s.chainedA = w.carry.chainedA;

   for (int t = 0; t < 4; t++)
      cold.tim[t] = w.carry.tim[t];

   cold.emitClock = w.carry.emitClock;
   cold.emitValid = w.carry.emitValid;
   cold.lastUnlock = s.clock;

   /* detector records as assumed (at rest unless the chain kernel knows better); the running sums start from zero */
   s.u.search = w.carry.search;

   for (int i = 0; i < 3; i++)
      s.u.search.detA[i].acc = 0.0f;
   s.u.search.detV.acc = 0.0f;

   for (int i = 0; i < 2; i++)
   {
      s.u.search.detF[i].acc = 0.0f;
      s.u.search.detF[i].pulses = w.carry.pulsesF[i];
      s.u.search.detF[i].thr = w.carry.thrF[i];
   }

   /* every ring position starts at zero (the detector records are at rest, the rings are rebuilt by the warm-up) */
   cold.label[0] = nfc_label(s.clock, c.a[0].delay, c.a[0].p1, 0);
   cold.label[1] = nfc_label(s.clock, c.a[1].delay, c.a[1].p1, 0);
   cold.label[2] = nfc_label(s.clock, c.a[2].delay, c.a[2].p1, 0);
   cold.label[3] = nfc_label(s.clock, c.f[1].delay, c.f[1].p1, 0);
   cold.label[4] = nfc_label(s.clock, c.f[2].delay, c.f[2].p1, 0);
   cold.label[5] = nfc_label(s.clock, c.v.delay, c.v.p1, 0);
   cold.label[6] = nfc_label(s.clock, c.v.delay, c.v.p0, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* handing over between lanes                                                                  */
/* ------------------------------------------------------------------------------------------ */

#ifndef NFC_FENCE
#error "define NFC_FENCE() (device-wide memory fence) before including nfc_scan.hpp"
#endif

/* Can this lane's state be compared with another lane's at the same clock? Searching, and the detector bank has been
 * stepped without a break for NFC_WINDOW_STEADY samples (no lock, no sample below the power threshold): then every
 * history and correlation ring entry a detector can still read was written during that run and is a function of the
 * samples alone (up to the constant offset of the box sums, which never shows: nfc_step_upkeep). */
NFC_DEV bool nfc_lane_comparable(const NfcStreamState &s, const NfcStreamCold &cold)
{
   return s.lockTech == 0 && s.unlock == 0 && s.bankClock == s.clock && (uint32_t)(s.clock - cold.bankRun) >= NFC_WINDOW_STEADY;
}

NFC_DEV void nfc_mix(uint32_t h[2], uint32_t v)
{
   h[0] = (h[0] ^ v) * 0x01000193u;
   h[1] = (h[1] + v) * 0x9E3779B1u + (h[1] >> 15);
}

/* What decides the future of a searching decoder apart from the samples: the detector records (without the running
 * sums, whose offset never shows, and without what travels as NfcCarry) and the front end (equal anyway: scanned). */
NFC_DEV void nfc_lane_digest(const NfcStreamState &s, uint32_t h[2])
{
   const NfcSearchRegs &r = s.u.search;

   h[0] = 0x811C9DC5u;
   h[1] = 0x7F4A7C15u;

   nfc_mix(h, nfc_bits(s.env)); nfc_mix(h, nfc_bits(s.n1)); nfc_mix(h, nfc_bits(s.mdev)); nfc_mix(h, nfc_bits(s.avg));
   nfc_mix(h, nfc_bits(s.edgePeak)); nfc_mix(h, s.pulseFilter);

   (void)r; /* the detector records are compared as part of the carry (NfcCarry::search) */
}

/* a lane has reached its `verify` sample */
NFC_DEV void nfc_lane_publish(NfcWindow &w, const NfcStreamState &s, const NfcStreamCold &cold)
{
   if (nfc_lane_comparable(s, cold))
   {
      uint32_t h[2];
      nfc_lane_digest(s, h);
      w.pubDigest[0] = h[0];
      w.pubDigest[1] = h[1];
      nfc_carry_take(w.pubCarry, s, cold);
      w.pubTail = cold.frameTail;
      w.pubState = 1u;
   }
   else
      w.pubState = 2u;
}

/* A lane at sample `pos` (a tile boundary): is this the sample a later lane of the same stream publishes its state for,
 * and is this lane in a state that can be compared? Then it stops here and leaves its digest; whether the two states
 * were the same (the later lane then decodes the rest exactly as this one would have) is settled by the chain kernel,
 * which sends this lane on if they were not. The decision does not look at what the other lane has published so far:
 * which lane runs first must not change what a lane does. `succ` walks the job's windows [.., succEnd). */
NFC_DEV bool nfc_lane_handover(NfcWindow *windows, NfcWindow &me, uint32_t &succ, uint32_t succEnd, uint32_t pos, const NfcStreamState &s,
                               const NfcStreamCold &cold)
{
   while (succ < succEnd && (windows[succ].verify < pos || succ <= me.noHand))
      succ++;

#ifdef NFC_HANDOVER_TALLY
   if (succ < succEnd && windows[succ].verify == pos)
      NFC_HANDOVER_TALLY(s, cold, pos - me.start);
#endif
   if (succ >= succEnd || windows[succ].verify != pos || !nfc_lane_comparable(s, cold))
      return false;

   uint32_t h[2];
   nfc_lane_digest(s, h);

   me.handTo = succ;
   me.stopDigest[0] = h[0];
   me.stopDigest[1] = h[1];
   return true;
}

/* ------------------------------------------------------------------------------------------ */
/* chain: one thread per job, after every decode pass                                          */
/* ------------------------------------------------------------------------------------------ */

/* Follows the stream through its lanes: lane 0 carries the state in; a lane that retired at sample r hands over to the
 * first window that activates at or after r. A hand-over is sound when the window had assumed exactly what the lane
 * left (nfc_carry_same). Otherwise the window is marked to run again with what was left; from there on the walk goes by
 * prediction - a lane that left what it had assumed will pass through whatever it is given, any other lane leaves
 * what it left before, and every lane stops where it stopped before - so that one more pass usually settles the whole
 * stream. Predictions are only used to choose what to run; a job is done when a walk meets results only.
 * Returns true when the job needs another pass. lanes[] / colds[] are the virtual slots (index = window index). */
NFC_DEV bool nfc_chain_follow(NfcScanJob &job, uint32_t jobIndex, NfcWindow *windows, const NfcStreamState *lanes, const NfcStreamCold *colds,
                              uint32_t maxPasses)
{
   NfcWindow *w = windows + job.firstWindow; /* speculative windows, ordered by activation */
   const uint32_t n = job.windows;

   for (uint32_t i = 0; i < n; i++)
   {
      w[i].live = 0;
      w[i].liveFrom = 0;
      w[i].rerun = 0;
      w[i].pulsesFix[0] = w[i].pulsesFix[1] = 0;
      w[i].thrPass[0] = w[i].thrPass[1] = 0;
   }

   windows[jobIndex].rerun = 0;
   windows[jobIndex].pulsesFix[0] = windows[jobIndex].pulsesFix[1] = 0;
   windows[jobIndex].thrPass[0] = windows[jobIndex].thrPass[1] = 0;

   bool again = false;
   uint32_t lane = jobIndex; /* the carry lane comes first */
   uint32_t prev = jobIndex; /* the lane that handed over to `lane` */
   uint32_t next = 0;        /* first speculative window not yet passed */
   NfcCarry have;            /* what the stream's state holds where `lane` takes over */
   bool handed = false;      /* `lane` took over at its verify sample (else at its start) */

   windows[jobIndex].live = 1;
   windows[jobIndex].liveFrom = 0;

   for (;;)
   {
      NfcWindow &x = windows[lane];

      NfcCarry left;
      nfc_carry_take(left, lanes[lane], colds[lane]);

      if (lane != jobIndex)
      {
         /* what the lane had where it took over: its assumption at its start, or what it published */
         const NfcCarry &had = handed ? x.pubCarry : x.carry;

         /* protocol state of the technologies the lane never locked: it did not look at what it had assumed there and
          * hands on what the stream holds */
         const uint32_t used = colds[lane].usedTech;

         const uint32_t waits = colds[lane].waitFlags;

         for (int t = 0; t < 4; t++)
         {
            if (!((used >> t) & 1u))
               left.tim[t] = have.tim[t];
            else
            {
               if (!((used >> (8 + t)) & 1u))
                  left.tim[t].lastCommand = have.tim[t].lastCommand; /* never changed by the lane: what the stream holds */
               if (!((waits >> (4 + t)) & 1u))
                  left.tim[t].protoWaitingTime = have.tim[t].protoWaitingTime; /* (the same: a lane that runs on another value than the stream's either passes nfc_wait_admits below or runs again) */
            }
         }

         if (!(used & 1u))
            left.chainedA = have.chainedA;

         /* what an NFC-F detector remembers across its partial resets, when the lane neither looked at it nor started the
          * record over: unchanged, so it is what the stream holds */
         for (int i = 0; i < 2; i++)
         {
            if (!((used >> (12 + i)) & 1u) && !left.clearedF[i])
            {
               left.pulsesF[i] = have.pulsesF[i];
               left.thrF[i] = have.thrF[i];
            }

            /* the record itself, when the lane neither ran the tracker on it nor reset it */
            if (!((used >> (14 + i)) & 1u) && !((used >> (16 + i)) & 1u))
               left.search.detF[i] = have.search.detF[i];
         }

#ifdef NFC_CHAIN_STATS
         NFC_CHAIN_STATS(x.carry, have, used);
#endif
         /* kept for the finish: should this be the stream's last lane, what it never looked at is put right in the state
          * it leaves (nfc_final_fixup); a lane that has to run again gets its `want` below */
         x.want = have;

         /* an NFC-F pulse memory the lane did look at, but that was as good as the true one (nfc_fbound_admits): compared
          * as if it had been the true one; what the lane leaves is put right below */
         NfcCarry hadAdmitted = had;
         int32_t pulsesFix[2] = {0, 0};
         uint32_t thrPass[2] = {0u, 0u};

         for (int i = 0; i < 2; i++)
         {
            if (((used >> (12 + i)) & 1u) &&
                nfc_fbound_admits(colds[lane].boundF[i], had.pulsesF[i], had.thrF[i], have.pulsesF[i], have.thrF[i], handed && had.clearedF[i] != 0u,
                                  handed && had.ownF[i] != 0u))
            {
               pulsesFix[i] = (int32_t)(have.pulsesF[i] - had.pulsesF[i]);
               thrPass[i] = nfc_bits(have.thrF[i]) != nfc_bits(had.thrF[i]) ? 1u : 0u;
               hadAdmitted.pulsesF[i] = have.pulsesF[i];
               hadAdmitted.thrF[i] = have.thrF[i];
            }
         }

         /* The record of the last carrier frame (NfcStreamCold::emitClock / emitValid) is looked at when a carrier frame is
          * stamped - it says whether the decoder's edge time has been zeroed since the edge tracker last moved
          * (nfc_detect_carrier, nfc_wave_edge_time) - and nowhere else: a lane that has emitted no carrier frame did nothing
          * that depended on the record it was given. It is then held to the carrier state alone (on / off, compared below), and
          * hands on the record the stream really holds, with the edge time that goes with it where the lane ended. */
         if (!colds[lane].emitOwn)
         {
            hadAdmitted.emitValid = have.emitValid;
            hadAdmitted.emitClock = have.emitClock;
            hadAdmitted.edgeTime = have.edgeTime;

            left.emitValid = have.emitValid;
            left.emitClock = have.emitClock;
            left.edgeTime = nfc_edge_time(have, colds[lane].trackedEnd);
         }

         /* a protocol waiting time the lane did run on, but that was as good as the true one (nfc_wait_admits) */
         {
            const NfcStreamState &last = lanes[lane];
            const uint32_t lockedTech = last.lockTech ? last.lockTech - NFC_TECH_A : 4u;

            for (uint32_t t = 0; t < 4u; t++)
            {
               const bool midWait = lockedTech == t && ((waits >> t) & 1u) != 0u;

               if (((used >> t) & 1u) && nfc_wait_admits(colds[lane], t, had.tim[t].protoWaitingTime, have.tim[t].protoWaitingTime,
                                                          handed && ((had.waitOwn >> (4u + t)) & 1u) != 0u, midWait))
                  hadAdmitted.tim[t].protoWaitingTime = have.tim[t].protoWaitingTime;
            }
         }

         if (nfc_carry_same(hadAdmitted, have, handed, x.tracked, used))
         {
            for (int i = 0; i < 2; i++)
            {
               if (!left.clearedF[i])
                  left.pulsesF[i] += (uint32_t)pulsesFix[i];
               if (thrPass[i] && !left.ownF[i])
                  left.thrF[i] = have.thrF[i];

               x.pulsesFix[i] = left.clearedF[i] ? 0 : pulsesFix[i];
               x.thrPass[i] = (thrPass[i] && !left.ownF[i]) ? 1u : 0u;
            }
         }
         else
         {
            /* ran on a wrong assumption. At its start it has to assume `have`, corrected by what it did itself between
             * its start and the sample it took over at */
            NfcCarry want = have;
            if (handed)
            {
               want = x.carry;
               nfc_carry_predict(want, x.pubCarry, have); /* what it had not touched by then follows `have` */
            }
#ifdef NFC_CHAIN_TRACE
            NFC_CHAIN_TRACE(lane, had, have, left);
#endif
            if (handed && nfc_carry_same(want, x.carry, false, x.tracked))
            {
               /* nothing it could assume differently would make the two lanes agree at that sample (both were busy
                * with the same thing from different beginnings): the lane before has to go on past it. The walk goes
                * on from this lane all the same, on the guess that what it leaves does not depend on the difference:
                * whatever else is wrong further down is then put right in the same pass */
               /* (with what it assumed, unless the walk has already found that wrong: then with what it has been told to
                * assume - round 4 put the old assumption back here, and the lane ran a pass for the hand-over and another
                * for the assumption: the 545 000-sample lanes of config 5 three times instead of twice) */
               NfcWindow &before = windows[prev];
               before.noHand = lane;
               if (!before.rerun)
                  before.want = before.carry;
               before.rerun = 1;
               again = true;
            }
            else
            {
               /* again, and predict what it will leave */
               nfc_carry_predict(left, x.carry, want, used);
               x.want = want;
               x.rerun = 1;
               again = true;
            }
         }
      }

      x.live = 1;
      job.finalLane = lane;

      /* ran to the end of the submission in a state nothing can take over from (locked, or rings not yet steady): the
       * stream's final state is this lane's. One that got there in a comparable state hands over to the closing window */
      if (x.retired == 0 || (x.stop >= job.count && x.retired != 1u))
         break;

      have = left;

      if (x.retired == 2)
      {
         /* handed over at x.stop to a lane that had published the same state; after a rerun of that lane the two may no
          * longer agree: then this lane has to go on instead */
         NfcWindow &y = windows[x.handTo];

         if (y.pubState != 1u || y.pubDigest[0] != x.stopDigest[0] || y.pubDigest[1] != x.stopDigest[1] || y.verify != x.stop)
         {
            x.noHand = x.handTo;
            if (!x.rerun)
               x.want = x.carry; /* (a lane already told to assume something else keeps that: see above) */
            x.rerun = 1;
            again = true;

            if (y.pubState != 1u)
               break; /* that lane never reached a state to take over in: nothing after this lane is known */
         }

         prev = lane;
         lane = x.handTo;
         next = lane - job.firstWindow + 1;
         handed = true;
         y.liveFrom = y.pubTail ? y.pubTail : 0xFFFFFFFFu; /* 0xFFFFFFFF: from its first record */
         continue;
      }

      /* stopped at rest: the first window activating at or after that sample takes over (the closing window at the latest) */
      while (next < n && w[next].activate < x.stop)
         next++;

      if (next >= n)
         break;

      prev = lane;
      lane = job.firstWindow + next;
      next++;
      handed = false;
   }

   job.passes++;

   if (again && job.passes >= maxPasses)
   {
      job.status |= NFC_JOB_GIVEUP;
      again = false;
   }

   if (again)
      job.status |= NFC_JOB_RERUN;
   else
      job.status &= ~NFC_JOB_RERUN;

   return again;
}

/* ------------------------------------------------------------------------------------------ */
/* finish: one thread per job, once the chain is settled                                        */
/* ------------------------------------------------------------------------------------------ */

/* The state a stream's last lane leaves, where that lane ran on assumptions the chain let pass because the lane never
 * looked at them (NfcStreamCold::usedTech): those fields still hold the assumption; `have` is what the stream really held
 * where the lane took over (NfcWindow::want as nfc_chain_follow left it). The counterpart of the substitutions
 * nfc_chain_follow makes in `left` for the lanes in between. */
NFC_DEV void nfc_final_fixup(NfcStreamState &s, NfcStreamCold &cold, const NfcWindow &lane)
{
   const NfcCarry &have = lane.want;

   const uint32_t used = cold.usedTech;

   for (int t = 0; t < 4; t++)
   {
      if (!((used >> t) & 1u))
         cold.tim[t] = have.tim[t];
      else
      {
         if (!((used >> (8 + t)) & 1u))
            cold.tim[t].lastCommand = have.tim[t].lastCommand;
         if (!((cold.waitFlags >> (4 + t)) & 1u))
            cold.tim[t].protoWaitingTime = have.tim[t].protoWaitingTime; /* (nfc_wait_admits) */
      }
   }

   if (!(used & 1u))
      s.chainedA = have.chainedA;

   /* the record of the last carrier frame, when the lane has emitted none itself (nfc_chain_follow) */
   if (!cold.emitOwn)
   {
      cold.emitValid = have.emitValid;
      cold.emitClock = have.emitClock;
      s.edgeTime = nfc_edge_time(have, cold.trackedEnd);
   }

   /* the detector records are parked while a technology is locked */
   NfcSearchRegs &r = s.lockTech ? cold.parked : s.u.search;

   for (int i = 0; i < 2; i++)
   {
      const float acc = r.detF[i].acc; /* the lane's own running sum goes with the lane's rings */

      if (!((used >> (14 + i)) & 1u) && !((used >> (16 + i)) & 1u))
      {
         r.detF[i] = have.search.detF[i];
         r.detF[i].pulses = have.pulsesF[i];
         r.detF[i].thr = have.thrF[i];
      }
      else if (!((used >> (12 + i)) & 1u) && !cold.clearedF[i])
      {
         r.detF[i].pulses = have.pulsesF[i];
         r.detF[i].thr = have.thrF[i];
      }
      else
      {
         /* looked at, found as good as the true one (nfc_fbound_admits): counted on from the true one, and a threshold the
          * lane never set is the one the stream held */
         r.detF[i].pulses += (uint32_t)lane.pulsesFix[i];
         if (lane.thrPass[i])
            r.detF[i].thr = have.thrF[i];
      }

      r.detF[i].acc = acc;
   }
}

/* A lane of a windowed launch chains its frame records in the staging sink: [next][record as nfc_emit writes it];
 * NfcStreamCold::frameHead is the word offset + 1 of the lane's first record. Copies the records of the live lanes,
 * in stream order, to the frame sink under the stream's own id. */
NFC_DEV void nfc_finish_frames(const NfcScanJob &job, uint32_t jobIndex, const NfcWindow *windows, const NfcStreamCold *colds,
                               const uint32_t *staging, uint32_t *sink, uint32_t *sinkCtl, uint32_t sinkWords)
{
   for (uint32_t i = 0; i <= job.windows; i++)
   {
      const uint32_t lane = i == 0 ? jobIndex : job.firstWindow + i - 1u;

      if (!windows[lane].live)
         continue;

      /* a lane that took over at its verify sample: only what it chained after the record that was its last then */
      uint32_t at = colds[lane].frameHead;

      if (windows[lane].liveFrom != 0 && windows[lane].liveFrom != 0xFFFFFFFFu)
         at = staging[windows[lane].liveFrom - 1u];

      while (at)
      {
         const uint32_t *rec = staging + (at - 1u);
         const uint32_t len = rec[9] > NFC_STREAM_BYTES ? NFC_STREAM_BYTES : rec[9];
         const uint32_t words = NFC_FRAME_HEADER_WORDS + ((len + 3u) >> 2);

         const uint32_t to = NFC_ATOMIC_ADD(sinkCtl, words);

         if (to + NFC_FRAME_MAX_WORDS > sinkWords)
            NFC_ATOMIC_ADD(sinkCtl + 1, 1u);
         else
         {
            sink[to] = job.slot;
            for (uint32_t k = 1; k < words; k++)
               sink[to + k] = rec[1 + k];
         }

         at = rec[0];
      }
   }
}

#endif
