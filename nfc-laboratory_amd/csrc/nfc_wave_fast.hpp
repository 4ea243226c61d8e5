/*
 * nfc_wave_fast.hpp — the wave decoder's bulk paths (nfc_wave.hpp): for the stage the decoder is in, the wave forms for
 * every remaining sample of the tile what the step would form there - running sums, ring taps, correlations - and the
 * stage's gate: can this sample change anything but sums and ring entries? Samples up to the first gated one are
 * committed in bulk, that one is stepped (nfc_step_impl, the statement of what happens there).
 *
 * The gates restate, per stage, the early exits of the step functions of nfc_tech_*.hpp (which follow the reference's
 * loops: NfcA.cpp:217-411,812-1421, NfcB.cpp:238-432,684-1040, NfcF.cpp:206-408,641-1042, NfcV.cpp:236-435,672-1074); a
 * gate may be true more often than needed (the step then finds nothing to do), never less.
 *
 * Values, per kind of correlator:
 *   raw    box sums of the raw signal (search bank; NFC-A / NFC-V poll frames, all of NFC-F): on the capture grid
 *          (multiples of 2^-15, |x| <= 1, |sum| <= 128 + window) every partial sum is exact in fp32, so the running sum
 *          after each sample of the tile is the sum before the tile plus a wave prefix sum of (entering - leaving), bit
 *          for bit what the step's add-then-subtract leaves; off the grid the path is not taken.
 *   power  10 * filtered^2 over a window (NFC-A 106k and NFC-V listen frames), phase: 10 * filtered * filtered one symbol
 *          back (BPSK listen frames): not on a grid, so the running sum is walked sample by sample in the step's own order
 *          (one dependent add and subtract per sample, every lane the same walk), everything around it is per lane.
 *   Ring taps: the entry a sample's step reads was written either by an earlier sample of this tile (then it is that
 *   sample's sum, taken from LDS scratch) or before the tile (then the ring still holds it: no write of the tile can
 *   have touched that position yet).
 */
#ifndef NFC_AMD_WAVE_FAST_HPP
#define NFC_AMD_WAVE_FAST_HPP

#define NFC_FAST_GRID_BACK 640u /* samples on the grid behind a tile before the raw paths trust the history (472 + 64 + margin) */
#define NFC_FAST_SUM_LIMIT 128.0f

/* stage keys */
enum
{
   NFC_FK_NONE = 0,
   NFC_FK_SEARCH,
   NFC_FK_UPKEEP,
   NFC_FK_UNARMED,
   NFC_FK_A_POLL,
   NFC_FK_A_ASK_START,
   NFC_FK_A_ASK_SYMBOL,
   NFC_FK_A_BPSK_START,
   NFC_FK_A_BPSK_SYMBOL,
   NFC_FK_B_POLL,
   NFC_FK_B_START,
   NFC_FK_B_SYMBOL,
   NFC_FK_F_DATA,
   NFC_FK_F_START,
   NFC_FK_V_POLL,
   NFC_FK_V_START,
   NFC_FK_V_SYMBOL
};

/* per-lane values of the tile's sample this lane holds, valid from lane `from` on while the stage stays `key` */
struct NfcWaveFast
{
   uint32_t key;
   uint32_t from;    /* first lane the values are valid for */
   uint32_t clock0;  /* clock of the sample before the tile */
   uint32_t gridSince; /* clock from which on every sample has been on the grid */
   uint32_t gridValid;
   float c[6];       /* running sum after this lane's sample: search correlators A106 A212 A424 F212 F424 V; locked: c[0] */
   float s0[6], s1[6];
   float edge[2], deep[2]; /* NFC-B detectors: DC-removed signal and depth at their decode points */
};

NFC_DEV void nfc_wave_fast_begin(NfcWaveFast &f)
{
   f.key = NFC_FK_NONE;
   f.from = 0;
   f.clock0 = 0;
   f.gridSince = 0;
   f.gridValid = 0;
}

/* inclusive prefix sum over the lanes of the wave (exact: see the header) */
NFC_DEV float nfc_wave_scan_add(float v)
{
   const uint32_t lane = NFC_WAVE_LANE();

   for (uint32_t d = 1; d < NFC_LANES; d <<= 1)
   {
      const float t = NFC_WAVE_SHFL_UP_F(v, d);
      if (lane >= d)
         v += t;
   }

   return v;
}

/* maximum over the lanes of the wave, in every lane */
NFC_DEV float nfc_wave_max(float v)
{
   for (uint32_t d = 1; d < NFC_LANES; d <<= 1)
   {
      const float t = NFC_WAVE_SHFL_XOR_F(v, d);
      v = t > v ? t : v;
   }

   return v;
}

/* clock of the tile sample this lane holds */
NFC_DEV uint32_t nfc_wave_clock_of(const NfcWaveFast &f)
{
   return f.clock0 + 1u + NFC_WAVE_LANE();
}

NFC_DEV uint32_t nfc_wave_mod(uint32_t v, uint32_t p)
{
   return v % p;
}

/* One raw box-sum correlator over the tile's samples from lane `from` on.
 *   acc      running sum after the sample before lane `from`
 *   pos      ring position of that sample
 *   prevKnown  the ring entry one sample back is the running sum (bank stepped on the previous sample)
 *   writeFrom  first clock whose step writes the ring (NFC-F listen frames: one symbol before the guard ends)
 * Leaves the running sums in lds->sum[slot] (for the commit) and returns this lane's values. */
struct NfcWaveRaw
{
   float c, c2, c3;
};

NFC_DEV NfcWaveRaw nfc_wave_raw(NFC_WAVE_LDS NfcWaveLds *lds, uint32_t slot, const NfcWaveFast &f, uint32_t from, float acc, uint32_t delay, uint32_t w,
                                uint32_t p1, uint32_t shift, uint32_t base, uint32_t pos, bool prevKnown, uint32_t writeFrom)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const bool active = lane >= from;
   const uint32_t t = nfc_wave_clock_of(f);
   const uint32_t k = lane - from; /* samples after the first of the run */

   const float in = lds->ring[NFC_R_X + ((t - delay) & NFC_HMASK)];
   const float out = lds->ring[NFC_R_X + ((t - delay - w) & NFC_HMASK)];

   NfcWaveRaw r;
   r.c = acc + nfc_wave_scan_add(active ? in - out : 0.0f);

   NFC_WAVE_BARRIER();
   lds->sum[slot][lane] = r.c;
   NFC_WAVE_BARRIER();

   const uint32_t posj = nfc_wave_mod(pos + 1u + k, p1);

   /* the entry `shift` samples back: written by the tile if that sample belongs to the run and wrote the ring */
   const bool c2Here = active && k >= shift && (int32_t)(t - shift - writeFrom) >= 0;
   const float c2Ring = lds->ring[NFC_R_CORR + base + nfc_wave_mod(posj + p1 - shift, p1)];
   r.c2 = c2Here ? lds->sum[slot][active ? lane - (k >= shift ? shift : 0u) : lane] : c2Ring;

   const bool c3Here = active && k >= 1u && (int32_t)(t - 1u - writeFrom) >= 0;
   const float c3Ring = lds->ring[NFC_R_CORR + base + nfc_wave_mod(posj + p1 - 1u, p1)];
   r.c3 = c3Here ? lds->sum[slot][active && k >= 1u ? lane - 1u : lane] : ((k == 0u && prevKnown) ? acc : c3Ring);

   return r;
}

/* commit of a raw correlator: ring entries of the samples [from, from + run) (the last p1 of them), positions are the
 * caller's */
NFC_DEV void nfc_wave_raw_commit(NFC_WAVE_LDS NfcWaveLds *lds, const NfcWaveFast &f, uint32_t from, uint32_t run, float c, uint32_t p1, uint32_t base,
                                 uint32_t pos, uint32_t writeFrom)
{
   const uint32_t lane = NFC_WAVE_LANE();

   if (lane >= from && lane < from + run && lane + p1 >= from + run && (int32_t)(nfc_wave_clock_of(f) - writeFrom) >= 0)
      lds->ring[NFC_R_CORR + base + nfc_wave_mod(pos + 1u + (lane - from), p1)] = c;
}

/* running sum of a listen-mode integrator, walked in the step's order: sum += in[j]; sum -= out[j] for the samples from
 * `from` on whose clock has reached `integrateFrom`. Every lane walks (and leaves the same values in lds->sum[0]). */
NFC_DEV float nfc_wave_walk(NFC_WAVE_LDS NfcWaveLds *lds, const NfcWaveFast &f, uint32_t from, uint32_t n, float acc, float in, float out, uint32_t integrateFrom)
{
   const uint32_t lane = NFC_WAVE_LANE();

   NFC_WAVE_BARRIER();
   lds->sum[1][lane] = in;
   lds->sum[2][lane] = out;
   NFC_WAVE_BARRIER();

   for (uint32_t j = from; j < n; j++)
   {
      if ((int32_t)(f.clock0 + 1u + j - integrateFrom) >= 0)
      {
         acc += lds->sum[1][j];
         acc -= lds->sum[2][j];
      }
      lds->sum[0][j] = acc;
   }

   NFC_WAVE_BARRIER();
   return lds->sum[0][lane];
}

/* which stage, and can the bulk path be taken at all? */
NFC_DEV uint32_t nfc_wave_stage(const NfcConfig &c, const NfcStreamState &s, bool upkeep)
{
   if (upkeep)
      return NFC_FK_UPKEEP;

   if (s.unlock)
      return NFC_FK_NONE;

   if (s.lockTech == 0)
      return NFC_FK_SEARCH;

   const NfcDecodeRegs &d = s.u.decode;

   if (d.pendType)
      return NFC_FK_NONE;

   const bool poll = d.frameType == NFC_FRAME_POLL;
   const bool listen = d.frameType == NFC_FRAME_LISTEN;

   if (s.lockTech == NFC_TECH_A)
   {
      if (poll)
         return NFC_FK_A_POLL;
      if (listen && d.lockRate == 0)
         return d.frameStart ? NFC_FK_A_ASK_SYMBOL : NFC_FK_A_ASK_START;
      if (listen)
         return d.frameStart ? NFC_FK_A_BPSK_SYMBOL : NFC_FK_A_BPSK_START;
   }
   else if (s.lockTech == NFC_TECH_B)
   {
      if (poll)
         return NFC_FK_B_POLL;
      if (listen)
         return d.frameStart ? NFC_FK_B_SYMBOL : NFC_FK_B_START;
   }
   else if (s.lockTech == NFC_TECH_F)
   {
      if (poll || (listen && d.frameStart))
         return NFC_FK_F_DATA;
      if (listen)
         return NFC_FK_F_START;
   }
   else if (s.lockTech == NFC_TECH_V)
   {
      if (poll)
         return NFC_FK_V_POLL;
      if (listen)
         return d.frameStart ? NFC_FK_V_SYMBOL : NFC_FK_V_START;
   }

   return NFC_FK_NONE;
}

/* ---- search bank ---- */

/* gates of the eight detectors at this lane's sample (the early exits of nfc*_detect_rate) */
NFC_DEV uint32_t nfc_wave_search_gate(const NfcConfig &c, const NfcStreamState &s, const NfcWaveFast &f, const NfcWaveTile &tile)
{
   const uint32_t t = nfc_wave_clock_of(f);
   const NfcSearchRegs &r = s.u.search;
   uint32_t gate = 0; /* bit per detector: A106 A212 A424 B106 B212 F212 F424 V */

   if (c.enabled & 1u)
   {
      const float limit = tile.env * c.corrThreshold[0];

      for (int i = 0; i < 3; i++)
      {
         const NfcDetA &m = r.detA[i];
         const float num = f.s0[i] - f.s1[i];
         const bool timeout = m.peakTime && t > m.peakTime + c.a[i].p1;
         const bool eventful = t >= m.winStart && (nfc_may_exceed(num, (float)c.a[i].p2, limit) || t == m.winEnd);
         gate |= (timeout || eventful) ? 1u << i : 0u;
      }
   }

   if (c.enabled & 2u)
   {
      for (int i = 0; i < 2; i++)
      {
         const NfcDetB &m = r.detB[i];
         const float edge = f.edge[i], deep = f.deep[i];
         const bool reset = deep > c.maxDepth[1] || (m.auxTime && t > m.auxTime + c.b[i].p1);
         bool hit;

         if (!m.symStart)
            hit = edge < -(tile.env * c.minDepth[1]) || t == m.winEnd;
         else if (!m.symEnd)
            hit = t < m.winStart ? edge > m.thr : ((edge > m.thr && edge > m.aux) || t == m.winEnd);
         else
            hit = t < m.winStart ? edge < -m.thr : ((edge < -m.thr && m.aux > edge) || t == m.winEnd);

         gate |= (reset || hit) ? 8u << i : 0u;
      }
   }

   if (c.enabled & 4u)
   {
      const float limit = tile.env * c.corrThreshold[2];

      for (int i = 0; i < 2; i++)
      {
         const NfcDetF &m = r.detF[i];
         const NfcRate &rt = c.f[i + 1];
         const float num = f.s0[3 + i] - f.s1[3 + i];
         const bool reset = tile.depth > c.maxDepth[2] || (m.peakTime && t > m.peakTime + rt.p1);
         const bool eventful = t >= m.winStart && (nfc_may_exceed(num, (float)rt.p2, limit) || t == m.sync || t == m.winEnd);
         gate |= (reset || eventful) ? 32u << i : 0u;
      }
   }

   if (c.enabled & 8u)
   {
      const NfcDetV &m = r.detV;
      const float limit = tile.env * c.corrThreshold[3];
      const float num = f.s0[5]; /* c2 - sum */
      const bool timeout = m.peakTime && t > m.peakTime + c.v.p0;
      const bool eventful = t >= m.winStart && (nfc_may_exceed(num, (float)c.v.p2, limit) || t == m.winEnd);
      gate |= (timeout || eventful) ? 128u : 0u;
   }

   return gate;
}

/* values of the search bank for the tile's samples from `from` on */
NFC_DEV void nfc_wave_search_values(const NfcConfig &c, const NfcStreamState &s, NFC_WAVE_LDS NfcWaveLds *lds, NfcWaveFast &f, uint32_t from)
{
   const uint32_t t = nfc_wave_clock_of(f);
   const bool prevKnown = s.bankClock == s.clock;
   const NfcSearchRegs &r = s.u.search;

   for (int i = 0; i < 3; i++)
   {
      const NfcRate &rt = c.a[i];
      const NfcWaveRaw v = nfc_wave_raw(lds, (uint32_t)i, f, from, r.detA[i].acc, rt.delay, rt.p2, rt.p1, rt.p1 - rt.p2, c.corrOffset[i], s.posA[i], prevKnown, s.clock - 0x40000000u);
      f.c[i] = v.c;
      f.s0[i] = v.c - v.c2;
      f.s1[i] = v.c2 - v.c3;
   }

   for (int i = 0; i < 2; i++)
   {
      const NfcRate &rt = c.f[i + 1];
      const NfcWaveRaw v = nfc_wave_raw(lds, 3u + (uint32_t)i, f, from, r.detF[i].acc, rt.delay, rt.p2, rt.p1, rt.p1 - rt.p2, c.corrOffset[3 + i], s.posF[i], prevKnown, s.clock - 0x40000000u);
      f.c[3 + i] = v.c;
      f.s0[3 + i] = v.c - v.c2;
      f.s1[3 + i] = v.c2 - v.c3;
   }

   {
      const NfcRate &rt = c.v;
      const NfcWaveRaw v = nfc_wave_raw(lds, 5u, f, from, r.detV.acc, rt.delay, rt.p2, rt.p1, rt.p1 - rt.p2, c.corrOffset[5], s.posV1, prevKnown, s.clock - 0x40000000u);
      f.c[5] = v.c;
      f.s0[5] = v.c2 - v.c; /* nfcv_detect: num = c2 - sum */
      f.s1[5] = 0.0f;
   }

   for (int i = 0; i < 2; i++)
   {
      const uint32_t slot = (t - c.b[i].delay) & NFC_HMASK;
      f.edge[i] = lds->ring[NFC_R_FILT + slot];
      f.deep[i] = lds->ring[NFC_R_DEPTH + slot];
   }
}

/* ---- locked stages ---- */

NFC_DEV bool nfc_wave_locked_gate(const NfcConfig &c, const NfcStreamState &s, const NfcWaveFast &f, const NfcWaveTile &tile, uint32_t key)
{
   const uint32_t t = nfc_wave_clock_of(f);
   const NfcDecodeRegs &d = s.u.decode;
   const NfcMod &m = d.lock;
   const NfcRate &rt = d.rt;

   switch (key)
   {
      /* Symbol stages whose window only gathers - the largest correlation and where it was, the values at the
       * synchronisation sample - and decides when it ends: the gathering is folded into the commit (nfc_wave_fold),
       * only the deciding sample is stepped. */
      case NFC_FK_A_POLL:
      case NFC_FK_A_ASK_SYMBOL:
      case NFC_FK_F_DATA:
      case NFC_FK_V_SYMBOL:
         return t >= m.winStart && t == m.winEnd;

      case NFC_FK_A_ASK_START:
      {
         const float s0 = f.s0[0];
         if (t < d.guardEnd)
            return false;
         const bool track = !m.symStart ? (s0 > m.thr && s0 > m.peak) : (s0 < -m.thr && s0 < m.peak);
         return t == d.guardEnd || t > d.waitingEnd || tile.depth > c.minDepth[0] || track || t == m.winEnd;
      }

      case NFC_FK_A_BPSK_START:
      {
         const float phase = f.c[0];
         if (t < d.guardEnd)
            return false;
         /* a negative phase with nothing tracked finds nothing to reset (the "preamble" it measures is the clock itself,
          * far outside 3..4 etu once the stream is a few thousand samples old) */
         const bool drop = !m.symEnd && phase < 0.0f && ((m.symStart | m.winEnd) != 0u || t <= 4096u);
         return t == d.guardEnd || t > d.waitingEnd || tile.depth > c.minDepth[0] || phase > m.thr || drop || t == m.winEnd;
      }

      case NFC_FK_A_BPSK_SYMBOL:
      case NFC_FK_B_SYMBOL:
      {
         const float phase = f.c[0];
         const bool cross = !m.auxTime && ((phase > 0.0f && m.lastPhase < 0.0f) || (phase < 0.0f && m.lastPhase > 0.0f));
         return cross || t == m.sync;
      }

      case NFC_FK_B_POLL:
      {
         const float edge = nfc_abs(f.edge[0]);
         return (t > m.winStart && t < m.winEnd && edge > m.thr && m.aux < edge) || t == m.sync;
      }

      case NFC_FK_B_START:
      {
         const float phase = f.c[0];
         if (t < d.guardEnd)
            return false;
         if (t == d.guardEnd || t > d.waitingEnd || tile.depth > c.maxDepth[1])
            return true;
         if (t < m.winStart)
            return false;
         /* a phase that is not positive ends what is being tracked; with nothing tracked (and the clock beyond any TR1)
          * the stage logic resets a record that is already clear */
         const bool tracked = (m.stage | m.winStart | m.winEnd | m.symStart | m.symEnd) != 0u || t <= 4096u;
         return phase > m.thr || t == m.winEnd || (!(phase > 0.0f) && tracked);
      }

      case NFC_FK_F_START:
      {
         const float sd = nfc_abs(f.s0[0] - f.s1[0]) / (float)rt.p2;
         if (t < d.guardEnd)
            return false;
         if (t == d.guardEnd || t > d.waitingEnd)
            return true;
         return t >= m.winStart && ((sd >= m.thr && sd > m.peak) || t == m.sync || t == m.winEnd);
      }

      case NFC_FK_V_POLL:
      {
         const float s0 = f.s0[0] / (float)rt.p2; /* (c2 - sum) / p2 */
         return t >= m.winStart && ((s0 > m.thr && s0 > m.peak) || t == m.winEnd);
      }

      case NFC_FK_V_START:
      {
         const float s0 = f.s0[0]; /* c2 - sum */
         if (t < d.guardEnd)
            return false;
         if (t == d.guardEnd || t > d.waitingEnd || tile.depth > c.maxDepth[3])
            return true;
         return t >= m.winStart && ((s0 < -m.thr && s0 < m.peak) || (s0 > m.thr && s0 > m.peak) || t == m.winEnd);
      }

      default:
         return true;
   }
}

/* the product ring entry of this lane's sample is written ahead (a function of the samples alone, read only at
 * older clocks than it is written at: nfc_wave.hpp), then the entry leaving the window is read */
NFC_DEV float nfc_wave_product(NFC_WAVE_LDS NfcWaveLds *lds, const NfcWaveFast &f, uint32_t from, float value, uint32_t delay, uint32_t window)
{
   const uint32_t cur = nfc_wave_clock_of(f) - delay;

   NFC_WAVE_BARRIER();
   if (NFC_WAVE_LANE() >= from)
      lds->ring[NFC_R_PROD + (cur & NFC_PMASK)] = value;
   NFC_WAVE_BARRIER();

   return lds->ring[NFC_R_PROD + ((cur - window) & NFC_PMASK)];
}

NFC_DEV void nfc_wave_locked_values(const NfcConfig &c, const NfcStreamState &s, NFC_WAVE_LDS NfcWaveLds *lds, NfcWaveFast &f, uint32_t from, uint32_t n, uint32_t key)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t t = nfc_wave_clock_of(f);
   const NfcDecodeRegs &d = s.u.decode;
   const NfcMod &m = d.lock;
   const NfcRate &rt = d.rt;
   const uint32_t never = s.clock - 0x40000000u;

   switch (key)
   {
      case NFC_FK_A_POLL:
      case NFC_FK_F_DATA:
      case NFC_FK_F_START:
      {
         /* NFC-F listen frames: the box sum runs from the end of the poll frame, the ring only from one symbol before the
          * guard ends (nfcf_listen_start) */
         const uint32_t writeFrom = key == NFC_FK_F_START ? d.guardEnd - rt.p1 : never;
         const NfcWaveRaw v = nfc_wave_raw(lds, 0u, f, from, m.acc, rt.delay, rt.p2, rt.p1, rt.p1 - rt.p2, d.lockBase, d.lockPos, false, writeFrom);
         f.c[0] = v.c;
         f.s0[0] = v.c - v.c2;
         f.s1[0] = v.c2 - v.c3;
         break;
      }

      case NFC_FK_V_POLL:
      {
         const NfcWaveRaw v = nfc_wave_raw(lds, 0u, f, from, m.acc, rt.delay, rt.p2, rt.p1, rt.p1 - rt.p2, d.lockBase, s.posV1, false, never);
         f.c[0] = v.c;
         f.s0[0] = v.c2 - v.c;
         f.s1[0] = 0.0f;
         break;
      }

      case NFC_FK_A_ASK_START:
      case NFC_FK_A_ASK_SYMBOL:
      case NFC_FK_V_START:
      case NFC_FK_V_SYMBOL:
      {
         const bool v15693 = key == NFC_FK_V_START || key == NFC_FK_V_SYMBOL;
         const uint32_t window = v15693 ? rt.p1 : rt.p2;
         const uint32_t period = v15693 ? rt.p0 : rt.p1;
         const uint32_t pos = v15693 ? s.posV0 : d.lockPos;
         const uint32_t shift = period - window; /* NFC-V: the entry one symbol half back in the two-symbol ring */

         const float v = lds->ring[NFC_R_FILT + ((t - rt.delay) & NFC_HMASK)];
         const float sq = v * v * 10.0f;
         const float old = nfc_wave_product(lds, f, from, sq, rt.delay, window);
         const float sum = nfc_wave_walk(lds, f, from, n, m.acc, sq, old, never);

         const uint32_t k = lane - from;
         const bool active = lane >= from;
         const uint32_t posj = nfc_wave_mod(pos + 1u + k, period);
         const float c2Ring = lds->ring[NFC_R_CORR + d.lockBase + nfc_wave_mod(posj + period - shift, period)];
         const float c2 = (active && k >= shift) ? lds->sum[0][active && k >= shift ? lane - shift : lane] : c2Ring;
         const float c3Ring = lds->ring[NFC_R_CORR + d.lockBase + nfc_wave_mod(posj + period - 1u, period)];
         const float c3 = (active && k >= 1u) ? lds->sum[0][active && k >= 1u ? lane - 1u : lane] : c3Ring;

         f.c[0] = sum;

         if (v15693)
         {
            f.s0[0] = c2 - sum;
            f.s1[0] = 0.0f;
         }
         else
         {
            f.s0[0] = sum - c2;
            f.s1[0] = c2 - c3;
         }
         break;
      }

      case NFC_FK_A_BPSK_START:
      case NFC_FK_A_BPSK_SYMBOL:
      case NFC_FK_B_START:
      case NFC_FK_B_SYMBOL:
      {
         const float a = lds->ring[NFC_R_FILT + ((t - rt.delay) & NFC_HMASK)];
         const float b = lds->ring[NFC_R_FILT + ((t - rt.delay - rt.p1) & NFC_HMASK)];
         const float in = a * b * 10.0f;
         const float out = nfc_wave_product(lds, f, from, in, rt.delay, rt.p4);
         /* NFC-A integrates from the end of the guard time on (nfca_listen_bpsk_start returns before it until then) */
         f.c[0] = nfc_wave_walk(lds, f, from, n, m.phaseAcc, in, out, key == NFC_FK_A_BPSK_START ? d.guardEnd : never);
         break;
      }

      case NFC_FK_B_POLL:
      {
         const uint32_t slot = (t - rt.delay) & NFC_HMASK;
         f.edge[0] = lds->ring[NFC_R_FILT + slot];
         f.deep[0] = lds->ring[NFC_R_DEPTH + slot];
         break;
      }

      default:
         break;
   }
}

/* What the samples [from, from + run) of a gathering symbol stage leave in the locked record (none of them is the
 * window's last): nfca_poll_symbol, nfca_listen_ask_symbol, nfcf_data_symbol, nfcv_listen_symbol up to their
 * `clock != winEnd` exits. Per sample: a correlation above the running maximum (and the stage's threshold) becomes the
 * maximum and marks its time; the synchronisation sample's values are kept. Over a run: the maximum is the largest
 * candidate if that beats the maximum before the run, its time the first sample that reaches it (later equal values do
 * not replace it: the comparison is strict). Results in lds->sum[6][8..15] for the uniform part:
 *   [8] 1 when the maximum moved, [9] the maximum, [10] its clock, [11] s0 there;
 *   [12] 1 when the synchronisation sample was in the run, [13] correlation, [14] s0, [15] s1 there. */
NFC_DEV void nfc_wave_fold(NFC_WAVE_LDS NfcWaveLds *lds, const NfcStreamState &s, const NfcWaveFast &f, uint32_t key, uint32_t from, uint32_t run)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t t = nfc_wave_clock_of(f);
   const NfcMod &m = s.u.decode.lock;
   const NfcRate &rt = s.u.decode.rt;

   const bool in = lane >= from && lane < from + run && t >= m.winStart;

   float sd;
   bool cand;

   if (key == NFC_FK_A_POLL)
   {
      sd = nfc_abs(f.s0[0] - f.s1[0]) / (float)rt.p2;
      cand = in && sd > m.thr;
   }
   else if (key == NFC_FK_A_ASK_SYMBOL)
   {
      sd = nfc_abs(f.s0[0] - f.s1[0]);
      cand = in && sd > m.peak; /* (no threshold; NaN: never a candidate) */
   }
   else if (key == NFC_FK_F_DATA)
   {
      sd = nfc_abs(f.s0[0] - f.s1[0]) / (float)rt.p2;
      cand = in && sd > m.thr;
   }
   else
   {
      sd = nfc_abs(f.s0[0]);
      cand = in && sd > m.thr;
   }

   const float top = nfc_wave_max(cand ? sd : -3.0e38f);
   const bool moved = top > m.peak;
   const uint64_t at = NFC_WAVE_BALLOT(cand && sd == top);
   const uint64_t atSync = NFC_WAVE_BALLOT(in && t == m.sync);

   NFC_WAVE_BARRIER();

   if (lane == 0)
   {
      lds->sum[6][8] = (moved && at) ? 1.0f : 0.0f;
      lds->sum[6][12] = atSync ? 1.0f : 0.0f;
   }

   if (moved && at && lane == (uint32_t)__builtin_ctzll(at))
   {
      lds->sum[6][9] = sd;
      lds->sum[6][10] = __builtin_bit_cast(float, t);
      lds->sum[6][11] = f.s0[0];
   }

   if (atSync && lane == (uint32_t)__builtin_ctzll(atSync))
   {
      lds->sum[6][13] = sd;
      lds->sum[6][14] = f.s0[0];
      lds->sum[6][15] = f.s1[0];
   }

   NFC_WAVE_BARRIER();
}

/* Samples from u.at on (at most up to n) that are committed in bulk; u.at is advanced past them. 0: the sample at u.at
 * has to be stepped. Called by every lane. */
NFC_DEV uint32_t nfc_wave_fast(const NfcConfig &c, NfcWaveUni &u, const NfcLaneMem &mem, NFC_WAVE_LDS NfcWaveLds *lds, NfcWaveFast &f, const NfcWaveTile &tile,
                               uint32_t n, bool upkeep, const NfcWaveItem &it)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t from = u.at;
   const NfcStreamState &s = u.s;

   uint32_t key = nfc_wave_stage(c, s, upkeep);

   if (key == NFC_FK_NONE)
   {
      f.key = NFC_FK_NONE;
      return 0u;
   }

   const uint32_t t = nfc_wave_clock_of(f);

   /* search: the bank is only stepped on armed samples (nfc_search_detect); a run is all armed or all unarmed */
   const bool armed = t >= 1024u && !(tile.env < c.powerThreshold);

   if (key == NFC_FK_SEARCH)
   {
      const bool firstArmed = ((NFC_WAVE_BALLOT(armed) >> from) & 1ull) != 0ull;
      if (!firstArmed)
         key = NFC_FK_UNARMED;
   }

   /* the raw box sums are only order-independent on the grid */
   const bool raw = key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP || key == NFC_FK_A_POLL || key == NFC_FK_F_DATA || key == NFC_FK_F_START || key == NFC_FK_V_POLL;

   if (raw && (!f.gridValid || (uint32_t)(f.clock0 + 1u - f.gridSince) < NFC_FAST_GRID_BACK))
   {
      f.key = NFC_FK_NONE; /* (stepping goes on without the values being kept up) */
      return 0u;
   }

   if (raw)
   {
      /* the running sums stay far inside the range in which multiples of 2^-15 are exact */
      bool small;

      if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
      {
         const NfcSearchRegs &r = s.u.search;
         small = nfc_abs(r.detA[0].acc) <= NFC_FAST_SUM_LIMIT && nfc_abs(r.detA[1].acc) <= NFC_FAST_SUM_LIMIT && nfc_abs(r.detA[2].acc) <= NFC_FAST_SUM_LIMIT &&
                 nfc_abs(r.detF[0].acc) <= NFC_FAST_SUM_LIMIT && nfc_abs(r.detF[1].acc) <= NFC_FAST_SUM_LIMIT && nfc_abs(r.detV.acc) <= NFC_FAST_SUM_LIMIT;
      }
      else
         small = nfc_abs(s.u.decode.lock.acc) <= NFC_FAST_SUM_LIMIT;

      if (!small)
      {
         f.key = NFC_FK_NONE;
         return 0u;
      }
   }

   /* ---- values (kept while the stage lasts) ---- */
   if (f.key != key || from < f.from)
   {
      if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
         nfc_wave_search_values(c, s, lds, f, from);
      else if (key != NFC_FK_UNARMED)
         nfc_wave_locked_values(c, s, lds, f, from, n, key);

      f.key = key;
      f.from = from;
   }

   /* ---- gate ---- */
   bool gate;

   /* a carrier frame is due (NfcDecoder.cpp:472-523: search mode only) */
   const bool carrier = (tile.avg > c.highThreshold) ? !s.carrierOn : ((tile.avg < c.lowThreshold) && !s.carrierOff);

   uint32_t which = 0;

   if (key == NFC_FK_SEARCH)
   {
      which = nfc_wave_search_gate(c, s, f, tile);
      gate = !armed || carrier || which != 0u;
   }
   else if (key == NFC_FK_UNARMED)
      gate = armed || carrier;
   else if (key == NFC_FK_UPKEEP)
      gate = false;
   else
      gate = nfc_wave_locked_gate(c, s, f, tile, key);

   const uint64_t gated = NFC_WAVE_BALLOT(gate && lane >= from && lane < n) >> from;
   const uint32_t run = gated ? (uint32_t)__builtin_ctzll(gated) : n - from;

   if (run == 0u)
   {
#ifdef NFC_WAVE_COUNT_DETECTORS
      if (lane == from)
         for (uint32_t b = 0; b < 8u; b++)
            if ((which >> b) & 1u)
               NFC_WAVE_COUNT_DETECTORS(b);
#endif
      return 0u;
   }

   NFC_WAVE_COUNT(key, 0u, run);

   /* ---- commit: ring entries by the lanes of the run, the state by everybody ---- */
   const uint32_t last = from + run - 1u;
   const uint32_t never = s.clock - 0x40000000u;

   NFC_WAVE_BARRIER();

   if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
   {
      const bool all = key == NFC_FK_UPKEEP; /* the warm-up keeps every correlator up (nfc_step_upkeep) */

      if (all || (c.enabled & 1u))
         for (int i = 0; i < 3; i++)
            nfc_wave_raw_commit(lds, f, from, run, f.c[i], c.a[i].p1, c.corrOffset[i], s.posA[i], never);

      if (all || (c.enabled & 4u))
         for (int i = 0; i < 2; i++)
            nfc_wave_raw_commit(lds, f, from, run, f.c[3 + i], c.f[i + 1].p1, c.corrOffset[3 + i], s.posF[i], never);

      if (all || (c.enabled & 8u))
         nfc_wave_raw_commit(lds, f, from, run, f.c[5], c.v.p1, c.corrOffset[5], s.posV1, never);
   }
   else if (key == NFC_FK_A_POLL || key == NFC_FK_F_DATA || key == NFC_FK_F_START)
      nfc_wave_raw_commit(lds, f, from, run, f.c[0], s.u.decode.rt.p1, s.u.decode.lockBase, s.u.decode.lockPos, key == NFC_FK_F_START ? s.u.decode.guardEnd - s.u.decode.rt.p1 : never);
   else if (key == NFC_FK_V_POLL)
      nfc_wave_raw_commit(lds, f, from, run, f.c[0], s.u.decode.rt.p1, s.u.decode.lockBase, s.posV1, never);
   else if (key == NFC_FK_A_ASK_START || key == NFC_FK_A_ASK_SYMBOL)
      nfc_wave_raw_commit(lds, f, from, run, f.c[0], s.u.decode.rt.p1, s.u.decode.lockBase, s.u.decode.lockPos, never);
   else if (key == NFC_FK_V_START || key == NFC_FK_V_SYMBOL)
      nfc_wave_raw_commit(lds, f, from, run, f.c[0], s.u.decode.rt.p0, s.u.decode.lockBase, s.posV0, never);

   const bool gathers = key == NFC_FK_A_POLL || key == NFC_FK_A_ASK_SYMBOL || key == NFC_FK_F_DATA || key == NFC_FK_V_SYMBOL;

   if (gathers)
      nfc_wave_fold(lds, s, f, key, from, run);

   /* the sums after the last sample of the run, for everybody */
   NFC_WAVE_BARRIER();
   if (lane == last)
   {
      for (int i = 0; i < 6; i++)
         lds->sum[6][i] = f.c[i];
   }
   NFC_WAVE_BARRIER();

   NFC_WAVE_UNIFORM_BEGIN(u)
   {
      NfcStreamState &w = u.s;
      const uint32_t firstClock = w.clock + 1u;

      if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
      {
         const bool all = key == NFC_FK_UPKEEP;
         NfcSearchRegs &r = w.u.search;

         if (all || (c.enabled & 1u))
         {
            r.detA[0].acc = lds->sum[6][0];
            r.detA[1].acc = lds->sum[6][1];
            r.detA[2].acc = lds->sum[6][2];
         }
         if (all || (c.enabled & 4u))
         {
            r.detF[0].acc = lds->sum[6][3];
            r.detF[1].acc = lds->sum[6][4];
         }
         if (all || (c.enabled & 8u))
            r.detV.acc = lds->sum[6][5];

         /* the bank has stepped on every sample of the run (nfc_search_detect / nfc_step_upkeep) */
         if (w.bankClock != firstClock - 1u)
            mem.cold->bankRun = firstClock;
         w.bankClock = w.clock + run;
      }
      else if (key == NFC_FK_A_BPSK_START || key == NFC_FK_A_BPSK_SYMBOL || key == NFC_FK_B_START || key == NFC_FK_B_SYMBOL)
         w.u.decode.lock.phaseAcc = lds->sum[6][0];
      else if (key != NFC_FK_UNARMED && key != NFC_FK_B_POLL)
         w.u.decode.lock.acc = lds->sum[6][0];

      if (gathers)
      {
         NfcMod &m = w.u.decode.lock;

         if (lds->sum[6][8] != 0.0f)
         {
            const uint32_t when = __builtin_bit_cast(uint32_t, lds->sum[6][10]);
            m.peak = lds->sum[6][9];

            if (key == NFC_FK_V_SYMBOL)
            {
               /* nfcv_listen_symbol keeps the correlation's sign and the sample */
               m.c0 = lds->sum[6][11];
               m.c1 = -lds->sum[6][11];
               m.symEnd = when;
            }
            else
               m.peakTime = when;
         }

         if (lds->sum[6][12] != 0.0f && key != NFC_FK_V_SYMBOL)
         {
            if (key != NFC_FK_F_DATA)
               m.cD = lds->sum[6][13];
            m.c0 = lds->sum[6][14];
            m.c1 = lds->sum[6][15];
         }
      }

      if (w.lockTech)
         w.u.decode.lockPos = (w.u.decode.lockPos + run) % w.u.decode.rt.p1;

      nfc_wave_advance(c, w, run);
      w.clock += run;
      w.env = lds->env[last];
      w.avg = lds->avg[last];
      w.mdev = lds->ring[NFC_R_MDEV + (w.clock & NFC_HMASK)];

      u.at += run;
   }
   NFC_WAVE_UNIFORM_END(u)

   return run;
}

#endif
