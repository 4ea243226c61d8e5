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

#define NFC_ENVELOPE_GROUP 16u /* samples fetched together (a quarter of a tile) */

#ifndef NFC_ENVELOPE_BIG
#define NFC_ENVELOPE_BIG 3.0e38f /* (NFC_SCAN_BIG of nfc_scan.hpp) */
#endif

NFC_DEV uint32_t nfc_envelope_bits(float v)
{
   uint32_t u;
   __builtin_memcpy(&u, &v, 4);
   return u;
}

/* one listed chunk (NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE), c: etu / envW0 / envW1 of the stream's configuration */
NFC_DEV void nfc_envelope_rewalk(const NfcConfig &c, const NfcScanArgs &A, NfcScanChunk ch)
{
   ch.index &= ~(NFC_CHUNK_REPAIR | NFC_CHUNK_ENVELOPE);

   const NfcScanJob *job = A.jobs + ch.job;
   const uint32_t g = job->firstChunk + ch.index; /* seam / chunk record */
   const uint32_t count = job->count;
   const uint32_t L = A.params.chunkSamples;
   const uint32_t start = ch.index * L;
   const uint32_t end = start + L < count ? start + L : count;

   if (start >= end)
      return;

   const uint8_t *data = job->data;
   const uint32_t stride = A.stride;

   NfcScanSeam seam = A.seams[g];

   uint32_t clock = A.states[job->slot].clock + start;
   float env = seam.start.env;
   uint32_t pulseFilter = seam.start.pulseFilter;

   /* the point stored where the chunk begins carries the true start too */
   if ((start % NFC_SCAN_POINT) == 0)
   {
      NfcScanPoint &first = A.points[job->firstPoint + start / NFC_SCAN_POINT];
      first.env = seam.start.env;
      first.pulseFilter = seam.start.pulseFilter;
   }

   /* the group being walked and the one after it (fetched while this one is walked) */
   float now[NFC_ENVELOPE_GROUP], ahead[NFC_ENVELOPE_GROUP];

   for (uint32_t k = 0; k < NFC_ENVELOPE_GROUP; k++)
      now[k] = NFC_SAMPLE_AT(data, stride, start + k < end ? start + k : end - 1u);

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
            break;
         }

         stored.env = env;
         stored.pulseFilter = pulseFilter;
      }

      float lo = NFC_ENVELOPE_BIG, hi = -NFC_ENVELOPE_BIG;

      for (uint32_t q = 0; q < NFC_SCAN_TILE; q += NFC_ENVELOPE_GROUP)
      {
         /* the next group (clamped to the chunk: what lies beyond is fetched and not walked) */
         const uint32_t next = pos + q + NFC_ENVELOPE_GROUP;

         if (next + NFC_ENVELOPE_GROUP <= end)
         {
            for (uint32_t k = 0; k < NFC_ENVELOPE_GROUP; k++)
               ahead[k] = NFC_SAMPLE_AT(data, stride, next + k);
         }
         else
         {
            for (uint32_t k = 0; k < NFC_ENVELOPE_GROUP; k++)
               ahead[k] = NFC_SAMPLE_AT(data, stride, next + k < end ? next + k : end - 1u);
         }

         for (uint32_t k = 0; k < NFC_ENVELOPE_GROUP; k++)
         {
            if (q + k < n)
            {
               ++clock;
               ++pulseFilter;
               nfc_envelope_step(c, clock, pulseFilter, env, now[k]);
               lo = env < lo ? env : lo;
               hi = env > hi ? env : hi;
            }
         }

         for (uint32_t k = 0; k < NFC_ENVELOPE_GROUP; k++)
            now[k] = ahead[k];
      }

      NfcScanTile &stat = A.tileStats[job->firstTile + pos / NFC_SCAN_TILE];
      stat.envmin = lo;
      stat.envmax = hi;
      stat.bits |= NFC_TILE_REWALKED;

      seam.end.env = env;
      seam.end.pulseFilter = pulseFilter;
   }

   A.seams[g] = seam; /* (the first walk's record with the tracker's end put right, or untouched after a merge) */
}

#endif
