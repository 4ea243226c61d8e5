/*
 * nfc_envelope.hpp — the second walk of the envelope tracker alone, a lane per chunk (nfc_envelope_kernel).
 *
 * After the first round of second walks every seam of a submission that still fails fails in the envelope tracker alone
 * (NFC_CHUNK_ENVELOPE, nfc_seams_check): the tracker is the one recurrence of the front end that is not contractive
 * (nfc_envelope_step: while the signal is modulated it holds), so inside a long exchange a chunk inherits a wrong envelope
 * from the chunk before, one chunk of the chain per round. nfc_scan_kernel walks those chunks too (its envelope-only
 * branch, the statement this file follows line by line), but inside its row machinery - 64 row fetches, 64 transposed LDS
 * stores and a barrier per 64-sample step, built for 64 lanes walking the whole front end - it is no faster per sample
 * than the walk of the whole front end (0.27 us: a round of the headline costs 9 ms of it, a short capture is 10 - 20
 * rounds of a millisecond). Here a lane fetches the samples of its own chunk, a group ahead of the one it walks, and does
 * nothing per sample but the tracker (nfc_envelope_step, the decoder's own function: NfcTech.cpp:36-55) and the envelope
 * extremes of the tile.
 *
 * What it reads and writes is what the scan kernel's branch reads and writes: the chunk's seam record (start: as the seam
 * check set it; end: the tracker's two fields put right, or the first walk's end where the walk has met its trajectory),
 * the stored points' two fields, the tiles' envelope extremes and NFC_TILE_REWALKED.
 *
 * The includer defines NFC_DEV and NFC_SAMPLE_AT(data, stride, index) and has included nfc_core.hpp, nfc_scan.h and
 * nfc_scan_launch.h.
 */
#ifndef NFC_AMD_ENVELOPE_HPP
#define NFC_AMD_ENVELOPE_HPP

#ifndef NFC_ENVELOPE_BIG
#define NFC_ENVELOPE_BIG 3.0e38f /* (NFC_SCAN_BIG of nfc_scan.hpp) */
#endif

NFC_DEV uint32_t nfc_envelope_bits(float v)
{
   uint32_t u;
   __builtin_memcpy(&u, &v, 4);
   return u;
}

/* One listed chunk (NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE) and the chain behind it; c: etu / envW0 / envW1 of the stream's
 * configuration. Round 5: the walk does not end with its chunk. Where it arrives at the next chunk with another envelope
 * than that chunk starts from, it goes on into it - unless somebody else walks that chunk this round (nfc_seams_check:
 * NFC_ZONE_LISTED / NFC_ZONE_FOLLOWS; a chunk listed like the chunk before it is left to that one's walk) -, until it meets
 * the trajectory on record: a chain of chunks that inherit a wrong envelope from each other is settled by one walk in one
 * round - it used to be a round per chunk, ten for a short capture, seven for config 5. This
 * is the statement (one thread; the emulated runtime runs it); the GPU runs nfc_envelope_rewalk_wave below, the same walk by
 * a wavefront. */
NFC_DEV void nfc_envelope_rewalk(const NfcConfig &c, const NfcScanArgs &A, NfcScanChunk ch)
{
   ch.index &= ~(NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE);

   const NfcScanJob *job = A.jobs + ch.job;
   const uint32_t count = job->count;
   const uint32_t L = A.params.chunkSamples;
   const uint8_t *data = job->data;
   const uint32_t stride = A.stride;

   if (A.followChains && (A.seams[job->firstChunk + ch.index].start.zone & NFC_ZONE_FOLLOWS))
      return; /* (walked with the chunk before it) */

   float env = 0.0f;
   uint32_t pulseFilter = 0;

   const uint32_t walkTo = A.followChains ? job->chunks : ch.index + 1u;

   for (uint32_t index = ch.index; index < walkTo; index++)
   {
      const uint32_t g = job->firstChunk + index; /* seam / chunk record */
      const uint32_t start = index * L;
      const uint32_t end = start + L < count ? start + L : count;

      if (start >= end)
         return;

      NfcScanSeam seam = A.seams[g];

      if (index != ch.index)
      {
         /* the next chunk of the chain: somebody else's this round (listed, and not as this one's follower), or one that starts
          * from the very envelope the walk arrives with (nothing to put right: the trajectory on record is the true one from
          * here), or one to go on into */
         const bool others = (seam.start.zone & NFC_ZONE_LISTED) != 0u && (seam.start.zone & NFC_ZONE_FOLLOWS) == 0u;
         const bool agrees = (seam.start.zone & NFC_ZONE_LISTED) == 0u && nfc_envelope_bits(seam.start.env) == nfc_envelope_bits(env) &&
                             seam.start.pulseFilter == pulseFilter;

         if (others || agrees)
            return;

         seam.start.env = env;
         seam.start.pulseFilter = pulseFilter;

         if (A.planesStale)
            A.planesStale[g] = 1u; /* (a start state rewritten: NfcScanArgs::planesStale) */
      }

      uint32_t clock = A.states[job->slot].clock + start;
      env = seam.start.env;
      pulseFilter = seam.start.pulseFilter;

      /* the point stored where the chunk begins carries the true start too */
      if ((start % NFC_SCAN_POINT) == 0)
      {
         NfcScanPoint &first = A.points[job->firstPoint + start / NFC_SCAN_POINT];
         first.env = seam.start.env;
         first.pulseFilter = seam.start.pulseFilter;
      }

      bool merged = false;

      for (uint32_t pos = start; pos < end; pos += NFC_SCAN_TILE)
      {
         const uint32_t n = end - pos < NFC_SCAN_TILE ? end - pos : NFC_SCAN_TILE;

         /* has the walk met the first one's trajectory? Then the rest of the chunk, and its end, stand as recorded */
         if (pos > start && (pos % NFC_SCAN_POINT) == 0)
         {
            NfcScanPoint &stored = A.points[job->firstPoint + pos / NFC_SCAN_POINT];

            if (nfc_envelope_bits(stored.env) == nfc_envelope_bits(env) && stored.pulseFilter == pulseFilter)
            {
               seam.end = A.seams[g].end;
               merged = true;
               break;
            }

            stored.env = env;
            stored.pulseFilter = pulseFilter;
         }

         float lo = NFC_ENVELOPE_BIG, hi = -NFC_ENVELOPE_BIG;

         for (uint32_t k = 0; k < n; k++)
         {
            ++clock;
            ++pulseFilter;
            nfc_envelope_step(c, clock, pulseFilter, env, NFC_SAMPLE_AT(data, stride, pos + k));
            lo = env < lo ? env : lo;
            hi = env > hi ? env : hi;
         }

         NfcScanTile &stat = A.tileStats[job->firstTile + pos / NFC_SCAN_TILE];
         stat.envmin = lo;
         stat.envmax = hi;
         stat.bits |= NFC_TILE_REWALKED;

         seam.end.env = env;
         seam.end.pulseFilter = pulseFilter;
      }

      A.seams[g] = seam; /* (the first walk's record with the tracker's start and end put right, or its end untouched after a merge) */

      if (merged)
      {
         /* from here on the walk would repeat the first one: the chunk ends as recorded */
         env = seam.end.env;
         pulseFilter = seam.end.pulseFilter;
      }
   }
}

#ifdef NFC_ENVELOPE_WAVE
/* The same walk by a whole wavefront (round 5; the form the GPU runs). The tracker is a recurrence: one lane's worth of
 * arithmetic per sample, whatever is done about it. What a lane per chunk cannot do is fetch its samples well - sixty-four
 * lanes of a wave read sixty-four chunks, a cache line each per load (profiles/r04/ab_envelope: slower than the scan kernel
 * for the long lists of a large submission) -, and what the scan kernel's row machinery cannot do is walk fast: 64 row
 * fetches, 64 transposed LDS stores and a barrier per 64-sample step (0.27 us per sample: a round costs the walk of one
 * chunk, 9 ms for the headline's 32768 samples, however few chunks are on its list). Here the wave is the chunk: lane j
 * loads sample j of the tile at hand (one coalesced load a tile, two tiles ahead, the magnitudes formed 64 at a time), then
 * every lane walks the same 64 steps on the samples handed round with v_readlane - the decoder's own nfc_envelope_step,
 * values every lane holds alike -, lane 0 writes what nfc_envelope_rewalk writes. A dozen dependent instructions per sample
 * instead of the row machinery's several hundred cycles; eight and more waves per SIMD fit.
 * NFC_ENVELOPE_READLANE_F(v, j): the value lane j holds. */
NFC_DEV void nfc_envelope_rewalk_wave(const NfcConfig &c, const NfcScanArgs &A, NfcScanChunk ch, uint32_t lane)
{
   ch.index &= ~(NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE);

   const NfcScanJob *job = A.jobs + ch.job;
   const uint32_t count = job->count;
   const uint32_t L = A.params.chunkSamples;
   const uint8_t *data = job->data;
   const uint32_t stride = A.stride;

   if (A.followChains && (A.seams[job->firstChunk + ch.index].start.zone & NFC_ZONE_FOLLOWS))
      return; /* (walked with the chunk before it) */

   float env = 0.0f;
   uint32_t pulseFilter = 0;

   const uint32_t walkTo = A.followChains ? job->chunks : ch.index + 1u;

   for (uint32_t index = ch.index; index < walkTo; index++)
   {
      const uint32_t g = job->firstChunk + index;
      const uint32_t start = index * L;
      const uint32_t end = start + L < count ? start + L : count;

      if (start >= end)
         return;

      NfcScanSeam seam = A.seams[g];

      if (index != ch.index)
      {
         const bool others = (seam.start.zone & NFC_ZONE_LISTED) != 0u && (seam.start.zone & NFC_ZONE_FOLLOWS) == 0u;
         const bool agrees = (seam.start.zone & NFC_ZONE_LISTED) == 0u && nfc_envelope_bits(seam.start.env) == nfc_envelope_bits(env) &&
                             seam.start.pulseFilter == pulseFilter;

         if (others || agrees)
            return;

         seam.start.env = env;
         seam.start.pulseFilter = pulseFilter;

         if (A.planesStale)
            A.planesStale[g] = 1u; /* (a start state rewritten: NfcScanArgs::planesStale) */
      }

      uint32_t clock = A.states[job->slot].clock + start;
      env = seam.start.env;
      pulseFilter = seam.start.pulseFilter;

      if ((start % NFC_SCAN_POINT) == 0 && lane == 0u)
      {
         NfcScanPoint &first = A.points[job->firstPoint + start / NFC_SCAN_POINT];
         first.env = seam.start.env;
         first.pulseFilter = seam.start.pulseFilter;
      }

      /* this lane's sample of the tile at hand and of the two after it (what lies beyond the chunk is fetched and not walked) */
      const uint32_t last = end - 1u;
      float x0 = NFC_SAMPLE_AT(data, stride, start + lane < last ? start + lane : last);
      float x1 = NFC_SAMPLE_AT(data, stride, start + NFC_SCAN_TILE + lane < last ? start + NFC_SCAN_TILE + lane : last);

      bool merged = false;

      for (uint32_t pos = start; pos < end; pos += NFC_SCAN_TILE)
      {
         const uint32_t n = end - pos < NFC_SCAN_TILE ? end - pos : NFC_SCAN_TILE;
         const uint32_t ahead = pos + 2u * NFC_SCAN_TILE + lane;
         const float x2 = NFC_SAMPLE_AT(data, stride, ahead < last ? ahead : last);

         if (pos > start && (pos % NFC_SCAN_POINT) == 0)
         {
            NfcScanPoint &stored = A.points[job->firstPoint + pos / NFC_SCAN_POINT];

            if (nfc_envelope_bits(stored.env) == nfc_envelope_bits(env) && stored.pulseFilter == pulseFilter)
            {
               seam.end = A.seams[g].end;
               merged = true;
               break;
            }

            if (lane == 0u)
            {
               stored.env = env;
               stored.pulseFilter = pulseFilter;
            }
         }

         float lo = NFC_ENVELOPE_BIG, hi = -NFC_ENVELOPE_BIG;
         bool walked = false;

         /* A whole tile past the stream's first symbol (where the tracker may still take the sample itself): the tracker's common
          * path, without a branch. nfc_envelope_step decides |x - env| / env < 0.05 without the division whenever the envelope is
          * positive and the deviation is clearly on one side of the limit (below 0.0499 env: yes; above 0.0501 env: no) - the
          * tile is walked on that assumption, a dozen vector instructions a sample with selects where the statement has
          * branches, and noting whether any sample was in neither case; only then (the envelope at zero, a ratio within 0.2 % of
          * the limit, a NaN) it is walked again by the statement itself. The values are every lane's alike, but written as
          * vector code on purpose: left to itself the compiler sees that they are uniform and turns every select into a scalar
          * branch on a vector comparison - four round trips between the two units per sample, ~1000 cycles (measured:
          * 14 ms per 32768-sample chunk, slower than the row machinery it was to replace). */
         if (n == NFC_SCAN_TILE && (uint32_t)(clock + 1u) >= (uint32_t)c.etu && (uint32_t)(clock + 1u + NFC_SCAN_TILE) > (uint32_t)(clock + 1u))
         {
            const uint32_t limit = (uint32_t)(c.etu * 10);
            const float w0 = c.envW0, w1 = c.envW1;

#ifndef NFC_ENVELOPE_NO_GROUPS
            /* Round 6: sixteen samples at a time, which of them the tracker follows decided for all sixteen at once. Inside a
             * group every update moves the envelope by less than 0.0501 w1 of itself (a sample it follows is within 5 % of it, and
             * the group is only taken when no sample can be followed for the counter's sake: counter + 16 <= ten symbols), so by
             * less than `drift` = 16 x 0.0506 w1 in all (0.8 % at this rate): a sample whose deviation from the envelope the
             * group begins with is below (0.0499 (1 - drift) - drift) of it is below 0.0499 of the envelope as it stands when the
             * sample's turn comes - the statement's own shortcut, nfc_envelope_step: followed -, one above (0.0501 (1 + drift) +
             * drift) is above 0.0501 of it - not followed. A group with a sample in neither case (the flank of an edge passing
             * through 4.1 - 5.9 %), an envelope that is not positive or a counter near its limit sends the whole tile to the walk
             * below, sample by sample. What is left of a group is e <- e a + b per sample with (a, b) = (w0, x w1) where the
             * sample is followed and (1, 0) where it is not - the same multiplication, the same addition, the same roundings as
             * the statement's e w0 + x w1 (and e 1 + 0 = e) -, two dependent operations instead of nine, and a group nothing of
             * which is followed (a pause, a modulated stretch) is a dozen instructions for its sixteen samples. */
            {
               const float drift = 16.0f * 0.0506f * w1;
               const float kBelow = 0.0499f * (1.0f - drift) - drift;
               const float kAbove = 0.0501f * (1.0f + drift) + drift;

               float e = env;
               uint32_t pf = pulseFilter;
               NFC_ENVELOPE_OPAQUE_F(e);
               float l = NFC_ENVELOPE_BIG, h = -NFC_ENVELOPE_BIG;
               bool ok = kBelow > 0.0f;
               const float xw = x0 * w1;

#pragma unroll
               for (uint32_t grp = 0; grp < NFC_SCAN_TILE / 16u; grp++)
               {
                  if (ok)
                  {
                     const bool inGroup = (lane >> 4) == grp;
                     const float dev = nfc_abs(x0 - e);
                     const bool below = dev < kBelow * e;
                     const bool above = dev > kAbove * e;
                     const uint64_t clear = NFC_ENVELOPE_BALLOT(inGroup && (below || above));
                     const uint64_t followed = NFC_ENVELOPE_BALLOT(inGroup && below);

                     ok = NFC_ENVELOPE_UNIFORM_POSITIVE(e) && clear == (0xFFFFull << (16u * grp)) && pf + 16u <= limit;

                     if (ok)
                     {
                        if (followed == 0ull)
                        {
                           /* (held throughout: the value the group's samples leave is the one it found) */
                           l = e < l ? e : l;
                           h = e > h ? e : h;
                           pf += 16u;
                        }
                        else
                        {
                           const float a = below ? w0 : 1.0f;
                           const float b = below ? xw : 0.0f;

#pragma unroll
                           for (uint32_t k = 0; k < 16u; k++)
                           {
                              const uint32_t j = 16u * grp + k;
                              e = e * NFC_ENVELOPE_READLANE_F(a, j) + NFC_ENVELOPE_READLANE_F(b, j);
                              l = e < l ? e : l;
                              h = e > h ? e : h;
                           }

                           /* the counter: samples since the last one followed */
                           const uint32_t lastFollowed = 63u - (uint32_t)__builtin_clzll(followed);
                           pf = 16u * grp + 15u - lastFollowed;
                        }
                     }
                  }
               }

               if (ok)
               {
#ifdef NFC_ENVELOPE_VERIFY
                  NFC_ENVELOPE_VERIFY(c, env, pulseFilter, x0, limit, e, pf, l, h);
#endif
                  env = e;
                  pulseFilter = pf;
                  clock += NFC_SCAN_TILE;
                  lo = l;
                  hi = h;
                  walked = true;
               }
            }

            if (!walked)
#endif
            {
            float e = env;
            uint32_t pf = pulseFilter;
            NFC_ENVELOPE_OPAQUE_F(e);
            NFC_ENVELOPE_OPAQUE_U(pf);
            bool rare = false;
            float l = NFC_ENVELOPE_BIG, h = -NFC_ENVELOPE_BIG;

#pragma unroll
            for (uint32_t j = 0; j < NFC_SCAN_TILE; j++)
            {
               const float x = NFC_ENVELOPE_READLANE_F(x0, j);
               const float dev = nfc_abs(x - e);
               const bool below = dev < 0.0499f * e;
               const bool above = dev > 0.0501f * e;
               rare = rare || !(e > 0.0f && (below || above));
               const uint32_t pf1 = pf + 1u;
               const bool update = below || pf1 > limit;
               const float followed = e * w0 + x * w1;
               e = update ? followed : e;
               pf = update ? 0u : pf1;
               l = e < l ? e : l;
               h = e > h ? e : h;
            }

            if (!NFC_ANY(rare))
            {
               env = e;
               pulseFilter = pf;
               clock += NFC_SCAN_TILE;
               lo = l;
               hi = h;
               walked = true;
            }
            }
         }

         if (!walked)
         {
            for (uint32_t j = 0; j < n; j++)
            {
               const float x = NFC_ENVELOPE_READLANE_F(x0, j);
               ++clock;
               ++pulseFilter;
               nfc_envelope_step(c, clock, pulseFilter, env, x);
               lo = env < lo ? env : lo;
               hi = env > hi ? env : hi;
            }
         }

         if (lane == 0u)
         {
            NfcScanTile &stat = A.tileStats[job->firstTile + pos / NFC_SCAN_TILE];
            stat.envmin = lo;
            stat.envmax = hi;
            stat.bits |= NFC_TILE_REWALKED;
         }

         seam.end.env = env;
         seam.end.pulseFilter = pulseFilter;

         x0 = x1;
         x1 = x2;
      }

      if (lane == 0u)
         A.seams[g] = seam;

      if (merged)
      {
         env = seam.end.env;
         pulseFilter = seam.end.pulseFilter;
      }
   }
}
#endif

#endif
