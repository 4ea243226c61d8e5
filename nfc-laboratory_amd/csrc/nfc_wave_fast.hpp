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
 *          for bit what the step's add-then-subtract leaves; off the grid (real radio input: float IQ) the
 *          sums are walked like the integrators below (nfc_wave_raw_walked).
 *   power  10 * filtered^2 over a window (NFC-A 106k and NFC-V listen frames), phase: 10 * filtered * filtered one symbol
 *          back (BPSK listen frames): not on a grid, so the running sum is walked sample by sample in the step's own order
 *          (one dependent add and subtract per sample, every lane the same walk), everything around it is per lane.
 *   Ring taps: the entry a sample's step reads was written either by an earlier sample of this tile (then it is that
 *   sample's sum, taken from LDS) or before the tile (then the ring still holds it: no write of the tile can have
 *   touched that position yet).
 * The per-lane values live in LDS (NfcWaveLds::sum / s0 / s1: element [k][lane] belongs to the tile sample lane `lane`
 * holds), valid from sample NfcWaveUni::from on while the stage stays NfcWaveUni::key: stepping a gated sample does not
 * change them (the step forms the same sums), a change of stage does.
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

/* the shared decoder state, read in place */
#define NFC_WAVE_STATE(lds) (*(const NfcStreamState *)&(lds)->u.s)

/* clock of the tile sample this lane holds */
NFC_DEV uint32_t nfc_wave_clock_of(uint32_t clock0)
{
   return clock0 + 1u + NFC_WAVE_LANE();
}

/* commit of a correlator ring: entries of the samples [from, from + run) (the last `period` of them) */
NFC_DEV void nfc_wave_ring_commit(NFC_WAVE_LDS NfcWaveLds *lds, uint32_t slot, uint32_t clock0, uint32_t from, uint32_t run, uint32_t period, uint32_t base,
                                  uint32_t pos, uint32_t writeFrom)
{
   const uint32_t lane = NFC_WAVE_LANE();

   if (lane >= from && lane < from + run && lane + period >= from + run && (int32_t)(nfc_wave_clock_of(clock0) - writeFrom) >= 0)
      lds->ring[NFC_R_CORR + base + nfc_wave_wrap3(pos + 1u + (lane - from), period)] = lds->sum[slot][lane];
}

/* The two ring taps of tile sample `j` of one correlator: the entry `shift` samples back and the entry one sample back.
 * Either was written by the tile if that sample belongs to the values formed (from `from` on) and wrote the ring - then
 * it is that sample's sum -, or stands in the ring as the tile found it: no write of the tile can have reached that
 * position before sample j itself has been dealt with (a ring slot is written again one period later).
 *   acc / pos    running sum and ring position of the sample before `from`
 *   prevKnown    the ring entry one sample back from `from` is that running sum (bank stepped on the previous sample)
 *   writeFrom    first clock whose step writes the ring (NFC-F listen frames: one symbol before the guard ends) */
NFC_DEV void nfc_wave_taps(NFC_WAVE_LDS NfcWaveLds *lds, uint32_t slot, uint32_t j, uint32_t clock0, uint32_t from, float acc, uint32_t p1, uint32_t shift,
                           uint32_t base, uint32_t pos, bool prevKnown, uint32_t writeFrom, float &c2, float &c3)
{
   const bool active = j >= from;
   const uint32_t t = clock0 + 1u + j;
   const uint32_t k = j - from; /* samples after the first of the run */
   const uint32_t posj = nfc_wave_wrap3(pos + 1u + (active ? k : 0u), p1);

   const bool c2Here = active && k >= shift && (int32_t)(t - shift - writeFrom) >= 0;
   const float c2Ring = lds->ring[NFC_R_CORR + base + nfc_wave_wrap1(posj + p1 - shift, p1)];
   c2 = c2Here ? lds->sum[slot][c2Here ? j - shift : j] : c2Ring;

   const bool c3Here = active && k >= 1u && (int32_t)(t - 1u - writeFrom) >= 0;
   const float c3Ring = lds->ring[NFC_R_CORR + base + nfc_wave_wrap1(posj + p1 - 1u, p1)];
   c3 = c3Here ? lds->sum[slot][c3Here ? j - 1u : j] : ((active && k == 0u && prevKnown) ? acc : c3Ring);
}

/* The two differences a detector or symbol stage looks at, at tile sample `j` (a lane's own sample, or the uniform
 * sample a step is about): S0 and S1 of the reference's correlators (NfcA.cpp:236-255, NfcF.cpp:247-262) - or, where a
 * stage only looks at one (NFC-V: NfcV.cpp:262-275), that one and zero - from the running sums of the tile
 * (NfcWaveLds::sum) and the ring, with the correlator as it stood when the sums were formed (NfcWaveUni::tap*).
 * key: the stage the values belong to; slot: search bank 0..2 NFC-A, 3..4 NFC-F, 5 NFC-V; locked stages 0. */
NFC_DEV void nfc_wave_s0s1(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t key, uint32_t slot, uint32_t j, float &s0, float &s1)
{
   const uint32_t clock0 = lds->u.clock0, from = lds->u.from;
   const float sum = lds->sum[slot][j];
   float c2, c3;

   if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
   {
      const bool prevKnown = lds->u.tapPrev != 0u;
      const uint32_t never = clock0 - 0x40000000u;
      const NfcRate &rt = slot < 3u ? c.a[slot] : (slot < 5u ? c.f[slot - 2u] : c.v);

      nfc_wave_taps(lds, slot, j, clock0, from, lds->u.tapAcc[slot], rt.p1, rt.p1 - rt.p2, c.corrOffset[slot], lds->u.tapPos[slot], prevKnown, never, c2, c3);

      s0 = slot == 5u ? c2 - sum : sum - c2; /* nfcv_detect: num = c2 - sum */
      s1 = slot == 5u ? 0.0f : c2 - c3;
      return;
   }

   nfc_wave_taps(lds, 0u, j, clock0, from, lds->u.tapAcc[0], lds->u.tapPeriod, lds->u.tapShift, lds->u.tapBase, lds->u.tapPos[0], false, lds->u.tapWriteFrom, c2, c3);

   const bool single = key == NFC_FK_V_POLL || key == NFC_FK_V_START || key == NFC_FK_V_SYMBOL;

   s0 = single ? c2 - sum : sum - c2;
   s1 = single ? 0.0f : c2 - c3;
}

/* search bank, NFC-A: the correlation of rate R at tile sample j (what nfca_detect_decide is given) */
NFC_DEV float nfc_wave_search_num(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t slot, uint32_t j)
{
   float s0, s1;
   nfc_wave_s0s1(c, lds, NFC_FK_SEARCH, slot, j, s0, s1);
   return s0 - s1;
}

/* Running sum of a listen-mode integrator, walked in the step's order: sum += in[j]; sum -= out[j] for the samples from
 * `from` on whose clock has reached `integrateFrom`. Every lane walks the same walk; the sums go to lds->sum[0]. */
NFC_DEV void nfc_wave_walk(NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t from, uint32_t n, float acc, float in, float out, uint32_t integrateFrom)
{
   const uint32_t lane = NFC_WAVE_LANE();

   NFC_WAVE_BARRIER();
   lds->sum[1][lane] = in;
   lds->sum[2][lane] = out;
   NFC_WAVE_BARRIER();

   float mine = acc;

   for (uint32_t j = from; j < n; j++)
   {
      if ((int32_t)(clock0 + 1u + j - integrateFrom) >= 0)
      {
         acc += NFC_WAVE_PICK_F(in, lds->sum[1], j);
         acc -= NFC_WAVE_PICK_F(out, lds->sum[2], j);
      }
      mine = lane == j ? acc : mine;
   }

   NFC_WAVE_BARRIER();
   lds->sum[0][lane] = mine;
   NFC_WAVE_BARRIER();
}

/* which stage? */
NFC_DEV uint32_t nfc_wave_stage(const NfcStreamState &s, bool upkeep)
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

/* The gates of the search bank, one detector at a time: does this lane's sample change the detector's record (given the
 * record as it stands - the state is that of the sample before the run)? Each restates the early exits of its detector
 * (nfca_detect_decide, nfcb_track, nfcf_detect_decide, nfcv_detect_decide). A gate that is up where nothing would have
 * changed only costs a step; one that is down where something would have is an error. */
/* (num: the correlation of this lane's sample, S0 - S1 - the taps of a tile's samples are formed once per call of nfc_wave_fast
 * and kept by the lanes: NfcSearchTaps) */
template <int I>
NFC_DEV bool nfc_wave_gate_a(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t t, float env, float num)
{
   /* nfca_detect_rate past its first exit: a correlation beyond the threshold only changes the record when it is a new
    * extreme of the pause being tracked (or, on the way down, a new deepest modulation) */
   const NfcDetA &m = NFC_WAVE_STATE(lds).u.search.detA[I];
   const NfcRate &rt = c.a[I];
   const float limit = env * c.corrThreshold[0];
   const bool timeout = m.peakTime && t > m.peakTime + rt.p1;
   bool moves = false;

   if (nfc_may_exceed(num, (float)rt.p2, limit))
   {
      const float sd = num / (float)rt.p2;
      const float deep = lds->ring[NFC_R_DEPTH + ((t - rt.delay - rt.p8) & NFC_FMASK)];
      moves = !m.symStart ? (sd < -limit && (sd < m.peak || deep > m.aux)) : (sd > limit && sd > m.peak);
   }

   return timeout || (t >= m.winStart && (moves || t == m.winEnd));
}

/* (the record `m` given: the bulk path steps this detector on its own, nfc_wave_fast; edge / deep: the DC-removed signal and
 * the modulation depth at the detector's decode point of this lane's sample) */
template <int I>
NFC_DEV bool nfc_wave_gate_b_at(const NfcConfig &c, const NfcDetB &m, uint32_t t, float env, float edge, float deep)
{
   const bool clear = (m.symStart | m.symEnd | m.winStart | m.winEnd | m.auxTime | nfc_bits(m.aux)) == 0u;
   const bool reset = (deep > c.maxDepth[1] || (m.auxTime && t > m.auxTime + c.b[I].p1)) && !clear;
   bool hit;

   if (!m.symStart)
      hit = (edge < -(env * c.minDepth[1]) && edge < m.aux) || t == m.winEnd;
   else if (!m.symEnd)
      hit = t < m.winStart ? edge > m.thr : ((edge > m.thr && edge > m.aux) || t == m.winEnd);
   else
      hit = t < m.winStart ? edge < -m.thr : ((edge < -m.thr && m.aux > edge) || t == m.winEnd);

   return reset || hit;
}

template <int I>
NFC_DEV bool nfc_wave_gate_b(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, const NfcDetB &m, uint32_t t, float env)
{
   const uint32_t slot = (t - c.b[I].delay) & NFC_FMASK;
   return nfc_wave_gate_b_at<I>(c, m, t, env, lds->ring[NFC_R_FILT + slot], lds->ring[NFC_R_DEPTH + slot]);
}

/* asked: the detector is told to reset; on a clear record that only leaves its mark (folded into the commit).
 * looked: the detector gets as far as looking at its record (a correlation that may exceed the threshold with the window
 * open): when the record is one the lane inherited, that leaves a mark too (NfcStreamCold::usedTech, bit 14 + I) */
template <int I>
NFC_DEV bool nfc_wave_gate_f(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t t, float env, float num, bool &asked, bool &looked)
{
   const NfcDetF &m = NFC_WAVE_STATE(lds).u.search.detF[I];
   const NfcRate &rt = c.f[I + 1];
   const float limit = env * c.corrThreshold[2];
   const float deep = lds->ring[NFC_R_DEPTH + (t & NFC_FMASK)];
   /* nfcf_detect_rate / nfcf_track_preamble: a correlation above the threshold only changes the record when it is the
    * largest of the pulse so far */
   const bool clear = (m.symStart | m.symEnd | m.winStart | m.winEnd | m.sync | m.peakTime | nfc_bits(m.peak)) == 0u;
   bool moves = false;

   asked = deep > c.maxDepth[2] || (m.peakTime && t > m.peakTime + rt.p1);
   looked = false;

   if (nfc_may_exceed(num, (float)rt.p2, limit))
   {
      const float sd = nfc_abs(num) / (float)rt.p2;
      moves = sd > limit && sd > m.peak;
      looked = t >= m.winStart;
   }

   return (asked && !clear) || (t >= m.winStart && (moves || t == m.sync || t == m.winEnd));
}

NFC_DEV bool nfc_wave_gate_v(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t t, float env, float num /* c2 - sum */)
{
   const NfcDetV &m = NFC_WAVE_STATE(lds).u.search.detV;
   const float limit = env * c.corrThreshold[3];
   /* nfcv_detect: a pulse correlation above the threshold only changes the record when it is the largest so far or comes
    * with a deeper modulation */
   const bool timeout = m.peakTime && t > m.peakTime + c.v.p0;
   bool moves = false;

   if (nfc_may_exceed(num, (float)c.v.p2, limit))
   {
      const float q = num / (float)c.v.p2;
      const float deep = nfc_wave_f_read(lds, NFC_R_DEPTH, t - c.v.delay - c.v.p8); /* (402 samples back: beyond the rings) */
      moves = q > limit && (q > m.peak || deep > m.aux);
   }

   return timeout || (t >= m.winStart && (moves || t == m.winEnd));
}

/* The gates of the bank, a word per sample of the tile (this lane's: NfcWaveLds::gate), kept while the records they were
 * evaluated for stand: bits 0..2 NFC-A, 3..4 NFC-B, 5..6 NFC-F, 7 NFC-V, 8..9 the NFC-F reset marks, 10..11 where the
 * NFC-F detectors look at their records (nfc_wave_gate_f). `valid`: the detectors whose bits still stand (a step clears
 * the bits of the detectors it asked). */
/* (bits 0 .. 7: the detectors are shown their records where their gates are up, in place: nfc_wave_fast) */
#define NFC_WAVE_GATE_OTHER 0x1000u /* bit 12: the bank is not armed at this sample, or a carrier frame is due */

/* What the six box-sum correlators of the search bank hold for this lane's sample of the tile: the differences the detectors
 * look at (nfc_wave_s0s1), formed once per call of nfc_wave_fast where the call first needs a correlator's - the first
 * evaluation of a gate, a detector shown its record - and kept by the lanes until the call returns (round 6: every gate
 * evaluated again after a detector had been shown its record formed its taps again, two ring reads and a dozen words of
 * the shared record a time: a seventh of the wave's cycles). They are functions of the tile's running sums and of the rings as
 * the tile found them: nothing a call does to the records changes them. The value at a sample the wave is about (uniform) is
 * the value the lane of that sample holds. */
struct NfcSearchTaps
{
   float numA[3];  /* NFC-A: S0 - S1 */
   float s0F[2];   /* NFC-F: S0 ... */
   float numF[2];  /* ... and S0 - S1 */
   float numV;     /* NFC-V: c2 - sum */
   uint32_t have;  /* bit per correlator (0..2 NFC-A, 3..4 NFC-F, 5 NFC-V): formed in this call (uniform) */
};

NFC_DEV void nfc_wave_search_tap(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, NfcSearchTaps &taps, uint32_t slot)
{
   if ((taps.have >> slot) & 1u)
      return;

   const uint32_t lane = NFC_WAVE_LANE();
   float s0, s1;
   nfc_wave_s0s1(c, lds, NFC_FK_SEARCH, slot, lane, s0, s1);

   if (slot < 3u)
      taps.numA[slot] = s0 - s1;
   else if (slot < 5u)
   {
      taps.s0F[slot - 3u] = s0;
      taps.numF[slot - 3u] = s0 - s1;
   }
   else
      taps.numV = s0;

   taps.have |= 1u << slot;
}

NFC_DEV uint32_t nfc_wave_search_bits(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t valid, uint32_t w, NfcSearchTaps &taps)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t t = nfc_wave_clock_of(clock0);
   const float env = lds->env[lane];

   if (c.enabled & 1u)
   {
      if (!(valid & 1u))
      {
         nfc_wave_search_tap(c, lds, taps, 0u);
         w = (w & ~1u) | (nfc_wave_gate_a<0>(c, lds, t, env, taps.numA[0]) ? 1u : 0u);
      }
      if (!(valid & 2u))
      {
         nfc_wave_search_tap(c, lds, taps, 1u);
         w = (w & ~2u) | (nfc_wave_gate_a<1>(c, lds, t, env, taps.numA[1]) ? 2u : 0u);
      }
      if (!(valid & 4u))
      {
         nfc_wave_search_tap(c, lds, taps, 2u);
         w = (w & ~4u) | (nfc_wave_gate_a<2>(c, lds, t, env, taps.numA[2]) ? 4u : 0u);
      }
   }

   if (c.enabled & 2u)
   {
      if (!(valid & 8u))
      {
         const NfcDetB m = NFC_WAVE_STATE(lds).u.search.detB[0];
         w = (w & ~8u) | (nfc_wave_gate_b<0>(c, lds, m, t, env) ? 8u : 0u);
      }
      if (!(valid & 16u))
      {
         const NfcDetB m = NFC_WAVE_STATE(lds).u.search.detB[1];
         w = (w & ~16u) | (nfc_wave_gate_b<1>(c, lds, m, t, env) ? 16u : 0u);
      }
   }

   if (c.enabled & 4u)
   {
      if (!(valid & 32u))
      {
         bool asked, looked;
         nfc_wave_search_tap(c, lds, taps, 3u);
         const bool gate = nfc_wave_gate_f<0>(c, lds, t, env, taps.numF[0], asked, looked);
         w = (w & ~0x520u) | (gate ? 0x20u : 0u) | (asked ? 0x100u : 0u) | (looked ? 0x400u : 0u);
      }
      if (!(valid & 64u))
      {
         bool asked, looked;
         nfc_wave_search_tap(c, lds, taps, 4u);
         const bool gate = nfc_wave_gate_f<1>(c, lds, t, env, taps.numF[1], asked, looked);
         w = (w & ~0xA40u) | (gate ? 0x40u : 0u) | (asked ? 0x200u : 0u) | (looked ? 0x800u : 0u);
      }
   }

   if (c.enabled & 8u)
   {
      if (!(valid & 128u))
      {
         nfc_wave_search_tap(c, lds, taps, 5u);
         w = (w & ~128u) | (nfc_wave_gate_v(c, lds, t, env, taps.numV) ? 128u : 0u);
      }
   }

   return w;
}

/* running sum of one raw box-sum correlator after each of the tile's samples from lane `from` on:
 *   acc    running sum after the sample before lane `from`
 *   delay  of the correlator's input, w: its window */
NFC_DEV float nfc_wave_raw_sum(NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t from, float acc, uint32_t delay, uint32_t w)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t t = nfc_wave_clock_of(clock0);

   const float in = lds->ring[NFC_R_X + ((t - delay) & NFC_HMASK)];
   const float out = lds->ring[nfc_wave_x_old_index(lds->ring, t - delay - w)];

   return acc + NFC_WAVE_SCAN_ADD_F(lane >= from ? in - out : 0.0f);
}

/* The same off the capture grid (or with a sum beyond the range in which grid values add exactly): the running sum walked
 * in the step's own order - sum += entering; sum -= leaving (nfc_corr_apply) -, every lane the same walk, a lane keeping
 * the sum after its own sample. n: samples of the tile. */
NFC_DEV float nfc_wave_raw_walked(NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t from, uint32_t n, float acc, uint32_t delay, uint32_t w)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t t = nfc_wave_clock_of(clock0);

   const float in = lds->ring[NFC_R_X + ((t - delay) & NFC_HMASK)];
   const float out = lds->ring[nfc_wave_x_old_index(lds->ring, t - delay - w)];

   float mine = acc;

   for (uint32_t j = from; j < n; j++)
   {
      acc += NFC_WAVE_SHFL_F(in, j);
      acc -= NFC_WAVE_SHFL_F(out, j);
      mine = lane == j ? acc : mine;
   }

   return mine;
}

NFC_DEV float nfc_wave_raw_any(NFC_WAVE_LDS NfcWaveLds *lds, bool walked, uint32_t clock0, uint32_t from, uint32_t n, float acc, uint32_t delay, uint32_t w)
{
   return walked ? nfc_wave_raw_walked(lds, clock0, from, n, acc, delay, w) : nfc_wave_raw_sum(lds, clock0, from, acc, delay, w);
}

/* values of the search bank for the tile's samples from `from` on: the six sums, then (one barrier) their taps */
NFC_DEV void nfc_wave_search_values(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t from, uint32_t n, bool walked)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const NfcStreamState &s = NFC_WAVE_STATE(lds);
   const bool prevKnown = s.bankClock == s.clock;
   const NfcSearchRegs &r = s.u.search;
   const uint32_t posA0 = s.posA[0], posA1 = s.posA[1], posA2 = s.posA[2], posF0 = s.posF[0], posF1 = s.posF[1], posV1 = s.posV1;
   const float accA0 = r.detA[0].acc, accA1 = r.detA[1].acc, accA2 = r.detA[2].acc, accF0 = r.detF[0].acc, accF1 = r.detF[1].acc, accV = r.detV.acc;

   const float sumA0 = nfc_wave_raw_any(lds, walked, clock0, from, n, accA0, c.a[0].delay, c.a[0].p2);
   const float sumA1 = nfc_wave_raw_any(lds, walked, clock0, from, n, accA1, c.a[1].delay, c.a[1].p2);
   const float sumA2 = nfc_wave_raw_any(lds, walked, clock0, from, n, accA2, c.a[2].delay, c.a[2].p2);
   const float sumF0 = nfc_wave_raw_any(lds, walked, clock0, from, n, accF0, c.f[1].delay, c.f[1].p2);
   const float sumF1 = nfc_wave_raw_any(lds, walked, clock0, from, n, accF1, c.f[2].delay, c.f[2].p2);
   const float sumV = nfc_wave_raw_any(lds, walked, clock0, from, n, accV, c.v.delay, c.v.p2);

   NFC_WAVE_BARRIER();
   lds->sum[0][lane] = sumA0;
   lds->sum[1][lane] = sumA1;
   lds->sum[2][lane] = sumA2;
   lds->sum[3][lane] = sumF0;
   lds->sum[4][lane] = sumF1;
   lds->sum[5][lane] = sumV;
   NFC_WAVE_BARRIER();

   /* the taps of a sample are formed where they are used (nfc_wave_s0s1), from the correlators as they stand now */
   NFC_WAVE_BARRIER();
   NFC_WAVE_UNIFORM_BEGIN
   {
      lds->u.tapPos[0] = posA0;
      lds->u.tapPos[1] = posA1;
      lds->u.tapPos[2] = posA2;
      lds->u.tapPos[3] = posF0;
      lds->u.tapPos[4] = posF1;
      lds->u.tapPos[5] = posV1;
      lds->u.tapAcc[0] = accA0;
      lds->u.tapAcc[1] = accA1;
      lds->u.tapAcc[2] = accA2;
      lds->u.tapAcc[3] = accF0;
      lds->u.tapAcc[4] = accF1;
      lds->u.tapAcc[5] = accV;
      lds->u.tapPrev = prevKnown ? 1u : 0u;
   }
   NFC_WAVE_UNIFORM_END
}

/* ---- locked stages ---- */

NFC_DEV bool nfc_wave_locked_gate(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t key)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t t = nfc_wave_clock_of(clock0);
   const NfcDecodeRegs &d = NFC_WAVE_STATE(lds).u.decode;
   const NfcMod &m = d.lock;
   const NfcRate &rt = d.rt;
   const float depth = lds->ring[NFC_R_DEPTH + (t & NFC_FMASK)]; /* of this lane's own sample */
   const float sum = lds->sum[0][lane];
   float s0 = 0.0f, s1 = 0.0f;

   /* (the stages that look at the correlator's differences) */
   if (key == NFC_FK_A_ASK_START || key == NFC_FK_F_START || key == NFC_FK_V_POLL || key == NFC_FK_V_START)
      nfc_wave_s0s1(c, lds, key, 0u, lane, s0, s1);

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
         if (t < d.guardEnd)
            return false;
         const bool track = !m.symStart ? (s0 > m.thr && s0 > m.peak) : (s0 < -m.thr && s0 < m.peak);
         return t == d.guardEnd || t > d.waitingEnd || depth > c.minDepth[0] || track || t == m.winEnd;
      }

      case NFC_FK_A_BPSK_START:
      {
         const float phase = sum;
         if (t < d.guardEnd)
            return false;
         /* a negative phase with nothing tracked finds nothing to reset (the "preamble" it measures is the clock itself,
          * far outside 3..4 etu once the stream is a few thousand samples old) */
         const bool drop = !m.symEnd && phase < 0.0f && ((m.symStart | m.winEnd) != 0u || t <= 4096u);
         return t == d.guardEnd || t > d.waitingEnd || depth > c.minDepth[0] || phase > m.thr || drop || t == m.winEnd;
      }

      case NFC_FK_A_BPSK_SYMBOL:
      case NFC_FK_B_SYMBOL:
      {
         const float phase = sum;
         const bool cross = !m.auxTime && ((phase > 0.0f && m.lastPhase < 0.0f) || (phase < 0.0f && m.lastPhase > 0.0f));
         return cross || t == m.sync;
      }

      case NFC_FK_B_POLL:
      {
         const float edge = nfc_abs(lds->ring[NFC_R_FILT + ((t - rt.delay) & NFC_FMASK)]);
         return (t > m.winStart && t < m.winEnd && edge > m.thr && m.aux < edge) || t == m.sync;
      }

      case NFC_FK_B_START:
      {
         const float phase = sum;
         if (t < d.guardEnd)
            return false;
         if (t == d.guardEnd || t > d.waitingEnd || depth > c.maxDepth[1])
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
         const float sd = nfc_abs(s0 - s1) / (float)rt.p2;
#ifdef NFC_WAVE_DEBUG_FSTART
         NFC_WAVE_DEBUG_FSTART(t, d, m, sd);
#endif
         if (t < d.guardEnd)
            return false;
         if (t == d.guardEnd || t > d.waitingEnd)
            return true;
         return t >= m.winStart && ((sd >= m.thr && sd > m.peak) || t == m.sync || t == m.winEnd);
      }

      case NFC_FK_V_POLL:
      {
         const float q = s0 / (float)rt.p2; /* (c2 - sum) / p2 */
         return t >= m.winStart && ((q > m.thr && q > m.peak) || t == m.winEnd);
      }

      case NFC_FK_V_START:
      {
         if (t < d.guardEnd)
            return false;
         if (t == d.guardEnd || t > d.waitingEnd || depth > c.maxDepth[3])
            return true;
         return t >= m.winStart && ((s0 < -m.thr && s0 < m.peak) || (s0 > m.thr && s0 > m.peak) || t == m.winEnd);
      }

      default:
         return true;
   }
}

/* the product ring entry of this lane's sample is written ahead (a function of the samples alone, read only at
 * older clocks than it is written at), then the entry leaving the window is read */
NFC_DEV float nfc_wave_product(NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t from, float value, uint32_t delay, uint32_t window)
{
   const uint32_t cur = nfc_wave_clock_of(clock0) - delay;

   NFC_WAVE_BARRIER();
   if (NFC_WAVE_LANE() >= from)
      lds->ring[NFC_R_PROD + (cur & NFC_PMASK)] = value;
   NFC_WAVE_BARRIER();

   return lds->ring[NFC_R_PROD + ((cur - window) & NFC_PMASK)];
}

NFC_DEV void nfc_wave_locked_values(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t from, uint32_t n, uint32_t key, bool walked)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t t = nfc_wave_clock_of(clock0);
   const NfcStreamState &s = NFC_WAVE_STATE(lds);
   const NfcDecodeRegs &d = s.u.decode;
   const NfcRate rt = d.rt;
   const uint32_t never = s.clock - 0x40000000u;
   const uint32_t lockBase = d.lockBase, lockPos = d.lockPos, guardEnd = d.guardEnd;
   const float acc = d.lock.acc, phaseAcc = d.lock.phaseAcc;
   const uint32_t posV0 = s.posV0, posV1 = s.posV1;

   /* the correlator as it stands, for the taps formed on demand (nfc_wave_s0s1) */
   uint32_t tapPeriod = rt.p1, tapShift = rt.p1 - rt.p2, tapPos = lockPos, tapWriteFrom = never;

   switch (key)
   {
      case NFC_FK_A_POLL:
      case NFC_FK_F_DATA:
      case NFC_FK_F_START:
      {
         /* NFC-F listen frames: the box sum runs from the end of the poll frame, the ring only from one symbol before the
          * guard ends (nfcf_listen_start) */
         tapWriteFrom = key == NFC_FK_F_START ? guardEnd - rt.p1 : never;

         const float sum = nfc_wave_raw_any(lds, walked, clock0, from, n, acc, rt.delay, rt.p2);

         NFC_WAVE_BARRIER();
         lds->sum[0][lane] = sum;
         NFC_WAVE_BARRIER();
         break;
      }

      case NFC_FK_V_POLL:
      {
         tapPos = posV1;

         const float sum = nfc_wave_raw_any(lds, walked, clock0, from, n, acc, rt.delay, rt.p2);

         NFC_WAVE_BARRIER();
         lds->sum[0][lane] = sum;
         NFC_WAVE_BARRIER();
         break;
      }

      case NFC_FK_A_ASK_START:
      case NFC_FK_A_ASK_SYMBOL:
      case NFC_FK_V_START:
      case NFC_FK_V_SYMBOL:
      {
         const bool v15693 = key == NFC_FK_V_START || key == NFC_FK_V_SYMBOL;
         const uint32_t window = v15693 ? rt.p1 : rt.p2;

         tapPeriod = v15693 ? rt.p0 : rt.p1;
         tapPos = v15693 ? posV0 : lockPos;
         tapShift = tapPeriod - window; /* NFC-V: the entry one symbol half back in the two-symbol ring */

         const float v = nfc_wave_f_read(lds, NFC_R_FILT, t - rt.delay); /* (NFC-V: 378 samples back, beyond the rings) */
         const float sq = v * v * 10.0f;
         const float old = nfc_wave_product(lds, clock0, from, sq, rt.delay, window);

         nfc_wave_walk(lds, clock0, from, n, acc, sq, old, never);
         break;
      }

      case NFC_FK_A_BPSK_START:
      case NFC_FK_A_BPSK_SYMBOL:
      case NFC_FK_B_START:
      case NFC_FK_B_SYMBOL:
      {
         const float a = lds->ring[NFC_R_FILT + ((t - rt.delay) & NFC_FMASK)];
         const float b = lds->ring[NFC_R_FILT + ((t - rt.delay - rt.p1) & NFC_FMASK)];
         const float in = a * b * 10.0f;
         const float out = nfc_wave_product(lds, clock0, from, in, rt.delay, rt.p4);
         /* NFC-A integrates from the end of the guard time on (nfca_listen_bpsk_start returns before it until then) */
         nfc_wave_walk(lds, clock0, from, n, phaseAcc, in, out, key == NFC_FK_A_BPSK_START ? guardEnd : never);
         break;
      }

      default:
         break;
   }

   NFC_WAVE_BARRIER();
   NFC_WAVE_UNIFORM_BEGIN
   {
      lds->u.tapPos[0] = tapPos;
      lds->u.tapAcc[0] = acc;
      lds->u.tapPeriod = tapPeriod;
      lds->u.tapShift = tapShift;
      lds->u.tapBase = lockBase;
      lds->u.tapWriteFrom = tapWriteFrom;
   }
   NFC_WAVE_UNIFORM_END
}

/* What the samples [from, from + run) of a gathering symbol stage leave in the locked record (none of them is the
 * window's last): nfca_poll_symbol, nfca_listen_ask_symbol, nfcf_data_symbol, nfcv_listen_symbol up to their
 * `clock != winEnd` exits. Per sample: a correlation above the running maximum (and the stage's threshold) becomes the
 * maximum and marks its time; the synchronisation sample's values are kept. Over a run: the maximum is the largest
 * candidate if that beats the maximum before the run, its time the first sample that reaches it (later equal values do
 * not replace it: the comparison is strict). Results in lds->u.pass[8..15] for the uniform part:
 *   [8] 1 when the maximum moved, [9] the maximum, [10] its clock, [11] s0 there;
 *   [12] 1 when the synchronisation sample was in the run, [13] correlation, [14] s0, [15] s1 there. */
NFC_DEV void nfc_wave_fold(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t clock0, uint32_t key, uint32_t from, uint32_t run)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const uint32_t t = nfc_wave_clock_of(clock0);
   const NfcDecodeRegs &d = NFC_WAVE_STATE(lds).u.decode;
   const float thr = d.lock.thr, peak = d.lock.peak;
   const uint32_t winStart = d.lock.winStart, sync = d.lock.sync;
   const float p2 = (float)d.rt.p2;
   float s0, s1;
   nfc_wave_s0s1(c, lds, key, 0u, lane, s0, s1);

   const bool in = lane >= from && lane < from + run && t >= winStart;

   float sd;
   bool cand;

   if (key == NFC_FK_A_POLL || key == NFC_FK_F_DATA)
   {
      sd = nfc_abs(s0 - s1) / p2;
      cand = in && sd > thr;
   }
   else if (key == NFC_FK_A_ASK_SYMBOL)
   {
      sd = nfc_abs(s0 - s1);
      cand = in && sd > peak; /* (no threshold; NaN: never a candidate) */
   }
   else
   {
      sd = nfc_abs(s0);
      cand = in && sd > thr;
   }

   const float top = NFC_WAVE_MAX_F(cand ? sd : -3.0e38f);
   const bool moved = top > peak;
   const uint64_t at = NFC_WAVE_BALLOT(cand && sd == top);
   const uint64_t atSync = NFC_WAVE_BALLOT(in && t == sync);

   NFC_WAVE_BARRIER();

   if (lane == 0)
   {
      lds->u.pass[8] = (moved && at) ? 1.0f : 0.0f;
      lds->u.pass[12] = atSync ? 1.0f : 0.0f;
   }

   if (moved && at && lane == (uint32_t)__builtin_ctzll(at))
   {
      lds->u.pass[9] = sd;
      lds->u.pass[10] = __builtin_bit_cast(float, t);
      lds->u.pass[11] = s0;
   }

   if (atSync && lane == (uint32_t)__builtin_ctzll(atSync))
   {
      lds->u.pass[13] = sd;
      lds->u.pass[14] = s0;
      lds->u.pass[15] = s1;
   }

   NFC_WAVE_BARRIER();
}

/* The marks the NFC-F detectors leave on samples [lo, end) of a run although nothing else happens there (bits 8 .. 11 of the
 * lanes' gate words: asked to reset a clear record; looked at a record the lane inherited before the lane has reset it,
 * NfcStreamCold::usedTech bits 16 + i / 14 + i), as nfcf_detect_rate leaves them sample after sample */
NFC_DEV uint32_t nfc_wave_f_marks(NFC_WAVE_LDS NfcWaveLds *lds, uint32_t bits, uint32_t lo, uint32_t end)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const bool mine = lane >= lo && lane < end;
   uint32_t marks = 0;

   for (uint32_t i = 0; i < 2u; i++)
   {
      const uint64_t asked = NFC_WAVE_BALLOT(mine && ((bits >> (8u + i)) & 1u)), looked = NFC_WAVE_BALLOT(mine && ((bits >> (10u + i)) & 1u));

      /* (a detector that looks at its record before the lane has reset it: NfcStreamCold::usedTech, bit 14 + i) */
      if (looked && !((lds->flags >> (16u + i)) & 1u) && (!asked || __builtin_ctzll(looked) < __builtin_ctzll(asked)))
         marks |= 1u << (14u + i);
      if (asked)
         marks |= 1u << (16u + i);
   }

   return marks;
}

/* what the step functions need of the lane's memory when they are only shown a detector record (no frame can be emitted) */
NFC_DEV NfcLaneMem nfc_wave_mem_records(NFC_WAVE_LDS NfcWaveLds *lds)
{
   NfcLaneMem mem;
   mem.ring = (NFC_RING_FLOAT *)lds->ring;
   mem.lane = 0;
   mem.exact = false;
   mem.linked = true;
   mem.flags = (uint32_t *)&lds->flags;
   mem.bytes = (uint8_t *)lds->bytes;
   mem.sink = nullptr;
   mem.sinkCursor = nullptr;
   mem.sinkDropped = nullptr;
   mem.sinkWords = 0;
   mem.streamId = 0;
   mem.cold = (NfcStreamCold *)&lds->cold;
   mem.tables = nullptr;
   mem.deep = &lds->deep;
   return mem;
}

/* Commits the samples from lds->u.at on that change nothing but sums and rings (at most up to n) and advances
 * lds->u.at past them; false: the sample at lds->u.at has to be stepped. Called by every lane. */
NFC_DEV bool nfc_wave_fast(const NfcConfig &c, NFC_WAVE_LDS NfcWaveLds *lds, uint32_t n, bool upkeep)
{
   const uint32_t lane = NFC_WAVE_LANE();
   const NfcStreamState &s = NFC_WAVE_STATE(lds);
   const uint32_t from = NFC_WAVE_UNIFORM_U32(lds->u.at);
   const uint32_t clock0 = NFC_WAVE_UNIFORM_U32(lds->u.clock0);

   NFC_WAVE_TICK(lds, 8u);
   uint32_t key = NFC_WAVE_UNIFORM_U32(nfc_wave_stage(s, upkeep));

   NFC_WAVE_COUNT(40u, 0u, 1u); /* calls */

   const uint32_t t = nfc_wave_clock_of(clock0);
   const float env = lds->env[lane];
   const float avg = lds->avg[lane];

   /* search: the bank is only stepped on armed samples (nfc_search_detect); a run is all armed or all unarmed */
   const bool armed = t >= 1024u && !(env < c.powerThreshold);

   if (key == NFC_FK_SEARCH)
   {
      const bool firstArmed = ((NFC_WAVE_BALLOT(armed) >> from) & 1ull) != 0ull;
      if (!firstArmed)
         key = NFC_FK_UNARMED;
   }

   /* The raw box sums are only order-independent on the grid, and while they stay far inside the range in which multiples
    * of 2^-15 are exact: there the running sums of a tile are a wave prefix sum away. Anywhere else - a sample off the
    * grid within the correlators' reach, a sum that has drifted out of that range - they are walked in the step's order
    * (nfc_wave_raw_walked). Looked at once per tile and stage: a sum moves by at most 2 per sample, and NFC_FAST_SUM_LIMIT
    * leaves room for a tile. */
   const bool raw = key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP || key == NFC_FK_A_POLL || key == NFC_FK_F_DATA || key == NFC_FK_F_START || key == NFC_FK_V_POLL;
   const bool take = key != NFC_FK_NONE;

   if (raw && NFC_WAVE_UNIFORM_U32(lds->u.takeKey) != key)
   {
      bool exact;

      /* (signed: a tile with a sample off the grid sets gridSince to its own end) */
      if ((int32_t)(clock0 + 1u - lds->u.gridSince) < (int32_t)NFC_FAST_GRID_BACK)
         exact = false;
      else if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
      {
         const NfcSearchRegs &r = s.u.search;
         exact = nfc_abs(r.detA[0].acc) <= NFC_FAST_SUM_LIMIT && nfc_abs(r.detA[1].acc) <= NFC_FAST_SUM_LIMIT && nfc_abs(r.detA[2].acc) <= NFC_FAST_SUM_LIMIT &&
                 nfc_abs(r.detF[0].acc) <= NFC_FAST_SUM_LIMIT && nfc_abs(r.detF[1].acc) <= NFC_FAST_SUM_LIMIT && nfc_abs(r.detV.acc) <= NFC_FAST_SUM_LIMIT;
      }
      else
         exact = nfc_abs(s.u.decode.lock.acc) <= NFC_FAST_SUM_LIMIT;

      NFC_WAVE_READ_FENCE();
      NFC_WAVE_UNIFORM_BEGIN
      {
         lds->u.takeKey = key;
         lds->u.walked = exact ? 0u : 1u;
      }
      NFC_WAVE_UNIFORM_END
   }

   const bool walked = raw && NFC_WAVE_UNIFORM_U32(lds->u.walked) != 0u;

   if (!take)
   {
      NFC_WAVE_COUNT(46u, 0u, 1u); /* bulk paths not taken */
      NFC_WAVE_COUNT(key, 1u, 0u);
#ifdef NFC_WAVE_COUNT_NOT_TAKEN
      NFC_WAVE_COUNT_NOT_TAKEN(key, false);
#endif
      /* stepping goes on without the values being kept up */
      NFC_WAVE_UNIFORM_BEGIN
      {
         lds->u.key = NFC_FK_NONE;
      }
      NFC_WAVE_UNIFORM_END
      return false;
   }

   /* ---- values (kept while the stage lasts) ---- */
   NFC_WAVE_TICK(lds, 2u);
   const bool formed = NFC_WAVE_UNIFORM_U32(lds->u.key) != key || from < NFC_WAVE_UNIFORM_U32(lds->u.from);

   if (formed)
   {
      if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
      {
         NFC_WAVE_COUNT(41u, 0u, 1u); /* search values formed */
         if (walked)
            NFC_WAVE_COUNT(48u, 0u, 1u); /* values with walked sums */
         nfc_wave_search_values(c, lds, clock0, from, n, walked);
      }
      else if (key != NFC_FK_UNARMED && key != NFC_FK_B_POLL)
      {
         NFC_WAVE_COUNT(42u, 0u, 1u); /* locked values formed */
         if (walked)
            NFC_WAVE_COUNT(48u, 0u, 1u);
         nfc_wave_locked_values(c, lds, clock0, from, n, key, walked);
      }

      NFC_WAVE_BARRIER();
      NFC_WAVE_UNIFORM_BEGIN
      {
         lds->u.key = key;
         lds->u.from = from;
      }
      NFC_WAVE_UNIFORM_END
   }

   /* ---- gate ---- */
   NFC_WAVE_TICK(lds, 3u);
   uint32_t which = 0;
   uint32_t run;
   uint64_t gated;   /* bit j: sample from + j is gated */
   uint32_t bits = 0; /* search: this lane's gate word */
   uint32_t marksFrom = from; /* search: first sample of the run whose NFC-F marks have not been left yet */

   /* a carrier frame is due (NfcDecoder.cpp:472-523: search mode only) */
   const bool carrier = (avg > c.highThreshold) ? !s.carrierOn : ((avg < c.lowThreshold) && !s.carrierOff);
   const uint64_t range = n < 64u ? (1ull << n) - 1ull : ~0ull;

   if (key == NFC_FK_SEARCH)
   {
      /* The detectors' gates are kept per sample while their records stand. A sample only NFC-B detectors still looking
       * for their first or second edge react to is taken in the run: it cannot lock, the records of the two rates are all
       * it touches (nfcb_track), and in a busy signal of the other technologies every pause is a falling edge to them. */
      NfcSearchTaps taps;
      taps.numA[0] = taps.numA[1] = taps.numA[2] = 0.0f;
      taps.s0F[0] = taps.s0F[1] = taps.numF[0] = taps.numF[1] = taps.numV = 0.0f;
      taps.have = 0u;

      bits = nfc_wave_search_bits(c, lds, clock0, formed ? 0u : NFC_WAVE_UNIFORM_U32(lds->u.maskValid), formed ? 0u : lds->gate[lane], taps);
      bits = (bits & ~NFC_WAVE_GATE_OTHER) | ((!armed || carrier) ? NFC_WAVE_GATE_OTHER : 0u);

      /* Only a sample at which the bank is not armed, or a carrier frame is due, is the step machine's from the start. Where a
       * detector's gate is up, the wave shows that detector its own record: its decision function (nfc*_detect_decide, the
       * statement of what the reference's detector does there) runs on a copy of the record, in the order of the bank (NFC-A
       * 106 / 212 / 424, NFC-B 106 / 212, NFC-F 212 / 424, NFC-V: NfcDecoder.cpp:401-417). Nearly always it moves the record and
       * nothing else - a pause being tracked, an edge, a pulse of a preamble -: the copies go back and the run goes on. When one
       * of them recognises its start of frame, the copies are dropped and the sample is the search step's (nfc_wave_search_step),
       * which decides again from the records as they were and locks; what the first decision has left outside the records - the
       * lane's marks and bounds of the NFC-F trackers, the decode set a locking detector prepares - it leaves again, the same. */
      const uint64_t hard = NFC_WAVE_BALLOT((bits & NFC_WAVE_GATE_OTHER) != 0u) & range;
      uint64_t soft = NFC_WAVE_BALLOT((bits & 0xFFu) != 0u) & range;
      uint32_t at = from, g = n;

      /* The NFC-B detectors alone (round 6). In a busy signal of the other technologies every pause is a falling edge to them -
       * two or three samples of a new minimum, the end of the window a quarter symbol later, the rising edge that clears the
       * record again: three samples in five at which the wave is brought to a record are theirs, two in five theirs alone -, and
       * all they touch is their own record of seven words (nfcb_track). For those samples the two records are kept in registers
       * from the first such sample of the call on, with what every lane's sample shows the detectors (the DC-removed signal and
       * the modulation depth at their decode points): the tracker - the decoder's own function - runs on the registers, the
       * gates of the rest of the tile are evaluated from them, the record goes back to the shared state without anybody
       * waiting for it. A visit used to be three round trips to LDS (the record, the two ring entries, the record again for
       * the gates) and the copying of a state for the detector to be shown. */
      bool heldB = false; /* (uniform: the registers below hold the records as they stand) */
      NfcDetB recB0, recB1;
      float edgeB0 = 0.0f, deepB0 = 0.0f, edgeB1 = 0.0f, deepB1 = 0.0f;
      __builtin_memset(&recB0, 0, sizeof(recB0));
      __builtin_memset(&recB1, 0, sizeof(recB1));

      for (;;)
      {
         const uint64_t rest = at < 64u ? (hard | soft) >> at : 0ull;

         g = rest ? at + (uint32_t)__builtin_ctzll(rest) : n;

         if (g >= n || ((hard >> g) & 1ull))
            break;

         const uint32_t here = NFC_WAVE_PICK_U32(bits, lds->gate, g);

#ifndef NFC_WAVE_NO_B_ALONE
         if ((here & 0xE7u) == 0u && (c.enabled & 2u))
         {
            if (!heldB)
            {
               const uint32_t slot0 = (t - c.b[0].delay) & NFC_FMASK, slot1 = (t - c.b[1].delay) & NFC_FMASK;

               recB0 = *(const NfcDetB *)&lds->u.s.u.search.detB[0];
               recB1 = *(const NfcDetB *)&lds->u.s.u.search.detB[1];
               edgeB0 = lds->ring[NFC_R_FILT + slot0];
               deepB0 = lds->ring[NFC_R_DEPTH + slot0];
               edgeB1 = lds->ring[NFC_R_FILT + slot1];
               deepB1 = lds->ring[NFC_R_DEPTH + slot1];
               heldB = true;
            }

            const uint32_t clk = clock0 + 1u + g;
            const float envAt = NFC_WAVE_SHFL_F(env, g);
            const float e0 = NFC_WAVE_SHFL_F(edgeB0, g), d0 = NFC_WAVE_SHFL_F(deepB0, g);
            const float e1 = NFC_WAVE_SHFL_F(edgeB1, g), d1 = NFC_WAVE_SHFL_F(deepB1, g);

            /* (the order of nfcb_detect: the second rate is not asked when the first has left the loop of the rates) */
            NfcDetB new0 = recB0, new1 = recB1;
            int r0 = 0, r1 = 0;

            if (here & 8u)
               r0 = nfcb_track<0>(c, new0, clk, envAt, e0, d0);
            if (r0 == 0 && (here & 16u))
               r1 = nfcb_track<1>(c, new1, clk, envAt, e1, d1);

            if (r0 == 1 || r1 == 1)
               break; /* a start of frame: the sample is the search step's (nothing has been moved) */

            NFC_WAVE_READ_FENCE(); /* (the records are about to change) */

            recB0 = new0;
            recB1 = new1;

            NFC_WAVE_UNIFORM_BEGIN
            {
               if (here & 8u)
                  *(NfcDetB *)&lds->u.s.u.search.detB[0] = new0;
               if (r0 == 0 && (here & 16u))
                  *(NfcDetB *)&lds->u.s.u.search.detB[1] = new1;
            }
            NFC_WAVE_UNIFORM_END

            NFC_WAVE_COUNT(44u, 0u, 1u); /* detectors shown their records in place */
#ifdef NFC_WAVE_COUNT_VISITS
            NFC_WAVE_COUNT(59u, 0u, 1u); /* NFC-B only */
#endif

            if (lane > g)
            {
               if (here & 8u)
                  bits = (bits & ~8u) | (nfc_wave_gate_b_at<0>(c, recB0, t, env, edgeB0, deepB0) ? 8u : 0u);
               if (here & 16u)
                  bits = (bits & ~16u) | (nfc_wave_gate_b_at<1>(c, recB1, t, env, edgeB1, deepB1) ? 16u : 0u);
            }

            soft = NFC_WAVE_BALLOT((bits & 0xFFu) != 0u) & range;
            at = g + 1u;
            continue;
         }

         /* (the visit below may move the NFC-B records where they lie) */
         if (here & 0x18u)
            heldB = false;
#endif
         const bool onF0 = (here & 0x20u) != 0u, onF1 = (here & 0x40u) != 0u;

         /* the marks of the samples before it (NFC-F: what its detectors leave where nothing else happens) are left first,
          * in the order the step machine would: the tracker below may reset a record */
         const uint32_t before = (onF0 || onF1) ? nfc_wave_f_marks(lds, bits, marksFrom, g) : 0u;

         /* the correlations at the sample the detectors are shown: what the lane of that sample holds (the taps of the detectors'
          * correlators formed now if the gates that brought the wave here were kept from an earlier call) */
         if (c.enabled & 1u)
         {
            if (here & 1u)
               nfc_wave_search_tap(c, lds, taps, 0u);
            if (here & 2u)
               nfc_wave_search_tap(c, lds, taps, 1u);
            if (here & 4u)
               nfc_wave_search_tap(c, lds, taps, 2u);
         }
         if ((c.enabled & 4u) && onF0)
            nfc_wave_search_tap(c, lds, taps, 3u);
         if ((c.enabled & 4u) && onF1)
            nfc_wave_search_tap(c, lds, taps, 4u);
         if ((c.enabled & 8u) && (here & 128u))
            nfc_wave_search_tap(c, lds, taps, 5u);

         const float numA0 = NFC_WAVE_SHFL_F(taps.numA[0], g), numA1 = NFC_WAVE_SHFL_F(taps.numA[1], g), numA2 = NFC_WAVE_SHFL_F(taps.numA[2], g);
         const float s0F0 = NFC_WAVE_SHFL_F(taps.s0F[0], g), s0F1 = NFC_WAVE_SHFL_F(taps.s0F[1], g);
         const float numF0 = NFC_WAVE_SHFL_F(taps.numF[0], g), numF1 = NFC_WAVE_SHFL_F(taps.numF[1], g);
         const float numV = NFC_WAVE_SHFL_F(taps.numV, g);

         NFC_WAVE_READ_FENCE(); /* (the records are about to change) */

         NFC_WAVE_TICK(lds, 10u);

         uint32_t lockedHere = 0u; /* (the verdict of the uniform block: every lane computes it on the GPU) */

         NFC_WAVE_UNIFORM_BEGIN
         {
            const uint32_t clk = clock0 + 1u + g;
            const float envAt = lds->env[g];
            const NfcLaneMem mem = nfc_wave_mem_records(lds);

            NfcStreamState shown; /* what the detectors are shown: the clock, the envelope and the records they are asked about */
            shown.clock = clk;
            shown.env = envAt;
            shown.posA[0] = shown.posA[1] = shown.posA[2] = 0u; /* (positions: only a lock looks at them) */
            shown.posF[0] = shown.posF[1] = 0u;
            shown.posV1 = shown.posV0 = 0u;

            bool locks = false;

            if (c.enabled & 1u)
            {
               const float limit = envAt * c.corrThreshold[0];

#define NFC_WAVE_A_ALONE(R, NUM)                                                                                                         \
               if (!locks && ((here >> R) & 1u))                                                                                         \
               {                                                                                                                         \
                  shown.u.search.detA[R] = *(const NfcDetA *)&lds->u.s.u.search.detA[R];                                                 \
                  locks = nfca_detect_decide<R>(c, shown, mem, NUM,                                                                      \
                                                lds->ring[NFC_R_DEPTH + ((clk - c.a[R].delay - c.a[R].p8) & NFC_FMASK)], limit, c.minDepth[0]); \
               }
               NFC_WAVE_A_ALONE(0, numA0)
               NFC_WAVE_A_ALONE(1, numA1)
               NFC_WAVE_A_ALONE(2, numA2)
#undef NFC_WAVE_A_ALONE
            }

            if (!locks && (c.enabled & 2u))
            {
               const uint32_t slot0 = (clk - c.b[0].delay) & NFC_FMASK, slot1 = (clk - c.b[1].delay) & NFC_FMASK;
               int r0 = 0;

               if ((here >> 3) & 1u)
               {
                  shown.u.search.detB[0] = *(const NfcDetB *)&lds->u.s.u.search.detB[0];
#ifdef NFC_WAVE_COUNT_B
                  NFC_WAVE_COUNT_B(c, shown.u.search.detB[0], 0, clk, envAt, lds->ring[NFC_R_FILT + slot0], lds->ring[NFC_R_DEPTH + slot0]);
#endif
                  r0 = nfcb_detect_decide<0>(c, shown, mem, lds->ring[NFC_R_FILT + slot0], lds->ring[NFC_R_DEPTH + slot0]);
               }

               if (r0 == 1)
                  locks = true;
               else if ((here >> 4) & 1u)
               {
                  /* (r0 == 2: the rate is skipped on this sample - its record is not looked at and goes back as it came) */
                  shown.u.search.detB[1] = *(const NfcDetB *)&lds->u.s.u.search.detB[1];
                  if (r0 == 0 && nfcb_detect_decide<1>(c, shown, mem, lds->ring[NFC_R_FILT + slot1], lds->ring[NFC_R_DEPTH + slot1]) == 1)
                     locks = true;
               }
            }

            if (!locks && (c.enabled & 4u) && (onF0 || onF1))
            {
               const float limit = envAt * c.corrThreshold[2];
               const float deep = lds->ring[NFC_R_DEPTH + (clk & NFC_FMASK)];

               lds->flags |= before;

               /* (what the search step leaves of the detectors it does not ask: nfc_wave_search_step) */
               lds->flags |= ((here >> 8) & 1u) << 16 | ((here >> 9) & 1u) << 17;
               const uint32_t seen = lds->flags;
               lds->flags = seen | (((here >> 10) & 1u & ~(seen >> 16)) << 14) | (((here >> 11) & 1u & ~(seen >> 17)) << 15);

               if (onF0)
               {
                  shown.u.search.detF[0] = *(const NfcDetF *)&lds->u.s.u.search.detF[0];
                  locks = nfcf_detect_decide<1>(c, shown, mem, s0F0, numF0, deep, limit);
               }

               if (!locks && onF1)
               {
                  shown.u.search.detF[1] = *(const NfcDetF *)&lds->u.s.u.search.detF[1];
                  locks = nfcf_detect_decide<2>(c, shown, mem, s0F1, numF1, deep, limit);
               }
            }

            if (!locks && (c.enabled & 8u) && ((here >> 7) & 1u))
            {
               shown.u.search.detV = *(const NfcDetV *)&lds->u.s.u.search.detV;
               locks = nfcv_detect_decide(c, shown, mem, numV, lds->ring[NFC_R_X + ((clk - c.v.delay) & NFC_HMASK)]);
            }

            /* every record that was shown, or none (the step decides about all of them again) */
            if (!locks)
            {
               if ((c.enabled & 1u) && (here & 1u))
                  *(NfcDetA *)&lds->u.s.u.search.detA[0] = shown.u.search.detA[0];
               if ((c.enabled & 1u) && (here & 2u))
                  *(NfcDetA *)&lds->u.s.u.search.detA[1] = shown.u.search.detA[1];
               if ((c.enabled & 1u) && (here & 4u))
                  *(NfcDetA *)&lds->u.s.u.search.detA[2] = shown.u.search.detA[2];
               if ((c.enabled & 2u) && (here & 8u))
                  *(NfcDetB *)&lds->u.s.u.search.detB[0] = shown.u.search.detB[0];
               if ((c.enabled & 2u) && (here & 16u))
                  *(NfcDetB *)&lds->u.s.u.search.detB[1] = shown.u.search.detB[1];
               if ((c.enabled & 4u) && onF0)
                  *(NfcDetF *)&lds->u.s.u.search.detF[0] = shown.u.search.detF[0];
               if ((c.enabled & 4u) && onF1)
                  *(NfcDetF *)&lds->u.s.u.search.detF[1] = shown.u.search.detF[1];
               if ((c.enabled & 8u) && (here & 128u))
                  *(NfcDetV *)&lds->u.s.u.search.detV = shown.u.search.detV;
            }

            lockedHere = locks ? 1u : 0u;
            NFC_WAVE_UNIFORM_LEAVE(lds->u.aloneLocked, lockedHere);
         }
         NFC_WAVE_UNIFORM_END

         NFC_WAVE_UNIFORM_TAKE(lds->u.aloneLocked, lockedHere);
#ifdef NFC_WAVE_PROFILE_VISITS
         NFC_WAVE_TICK(lds, 15u); /* (profile build: the gates after a visit on their own) */
#endif

         if (lockedHere)
            break; /* a start of frame: the sample is the search step's (nothing has been moved) */

         NFC_WAVE_COUNT(44u, 0u, 1u); /* detectors shown their records in place */
#ifdef NFC_WAVE_COUNT_VISITS
         for (uint32_t kk = 0; kk < 8u; kk++)
            if ((here >> kk) & 1u)
               NFC_WAVE_COUNT(51u + kk, 0u, 1u);
         if ((here & 0xFFu & ~0x18u) == 0u)
            NFC_WAVE_COUNT(59u, 0u, 1u); /* NFC-B only */
#ifdef NFC_WAVE_COUNT_RUN
         NFC_WAVE_COUNT_RUN(clock0 + 1u + g, here & 0xFFu);
#endif
#endif

         /* their gates over the rest of the tile, for the records as they now stand */
         if (lane > g)
         {
            if (c.enabled & 1u)
            {
               if (here & 1u)
                  bits = (bits & ~1u) | (nfc_wave_gate_a<0>(c, lds, t, env, taps.numA[0]) ? 1u : 0u);
               if (here & 2u)
                  bits = (bits & ~2u) | (nfc_wave_gate_a<1>(c, lds, t, env, taps.numA[1]) ? 2u : 0u);
               if (here & 4u)
                  bits = (bits & ~4u) | (nfc_wave_gate_a<2>(c, lds, t, env, taps.numA[2]) ? 4u : 0u);
            }
            if (c.enabled & 2u)
            {
               if (here & 8u)
               {
                  const NfcDetB m = s.u.search.detB[0];
                  bits = (bits & ~8u) | (nfc_wave_gate_b<0>(c, lds, m, t, env) ? 8u : 0u);
               }
               if (here & 16u)
               {
                  const NfcDetB m = s.u.search.detB[1];
                  bits = (bits & ~16u) | (nfc_wave_gate_b<1>(c, lds, m, t, env) ? 16u : 0u);
               }
            }
            if ((c.enabled & 4u) && onF0)
            {
               bool asked, looked;
               const bool gate = nfc_wave_gate_f<0>(c, lds, t, env, taps.numF[0], asked, looked);
               bits = (bits & ~0x520u) | (gate ? 0x20u : 0u) | (asked ? 0x100u : 0u) | (looked ? 0x400u : 0u);
            }
            if ((c.enabled & 4u) && onF1)
            {
               bool asked, looked;
               const bool gate = nfc_wave_gate_f<1>(c, lds, t, env, taps.numF[1], asked, looked);
               bits = (bits & ~0xA40u) | (gate ? 0x40u : 0u) | (asked ? 0x200u : 0u) | (looked ? 0x800u : 0u);
            }
            if ((c.enabled & 8u) && (here & 128u))
               bits = (bits & ~128u) | (nfc_wave_gate_v(c, lds, t, env, taps.numV) ? 128u : 0u);
         }

         if (onF0 || onF1)
            marksFrom = g + 1u;

         soft = NFC_WAVE_BALLOT((bits & 0xFFu) != 0u) & range;
         at = g + 1u;
      }

      NFC_WAVE_TICK(lds, 3u);
      run = g - from;
      /* (for the caller's wake rule the NFC-B gates do not count: the next call takes those samples) */
      gated = ((hard | (g < n ? 1ull << g : 0ull)) & ~((g < 64u ? 1ull << g : 0ull) - 1ull)) >> from;

      if (g < n)
         which = NFC_WAVE_PICK_U32(bits, lds->gate, g) & 0xFFFu;

      lds->gate[lane] = bits;

      NFC_WAVE_UNIFORM_BEGIN
      {
         lds->u.maskValid = 0xFFu;
      }
      NFC_WAVE_UNIFORM_END
   }
   else
   {
      bool gate;

      NFC_WAVE_TICK(lds, 9u);

      if (key == NFC_FK_UNARMED)
         gate = armed || carrier;
      else if (key == NFC_FK_UPKEEP)
         gate = false;
      else
         gate = nfc_wave_locked_gate(c, lds, clock0, key);

      gated = NFC_WAVE_BALLOT(gate && lane >= from && lane < n) >> from;
      run = gated ? (uint32_t)__builtin_ctzll(gated) : n - from;

      /* A waiting NFC-F decoder whose preamble tracker follows what the signal does (nfcf_listen_start between the guard
       * and the waiting time): after a pulse that did not hold the tracker starts over with a threshold of zero, and every
       * wiggle of the noise is a pulse to it for as long as the wait lasts (up to 217 k samples) - a step machine step per
       * sample. Such a sample touches the tracker's record and nothing else, so the wave applies the tracker itself
       * (nfcf_track_preamble, the decoder's own function, on a copy of the record) and goes on with the run, as it does for
       * the NFC-B detectors of the search bank; a preamble that completes is left to the step. */
      /* BPSK listen symbols (nfca_listen_bpsk_symbol, nfcb_listen_symbol): once per symbol the integrated phase changes sign,
       * which re-times the symbol - three fields of the record -; the decision at the synchronisation sample is the step's.
       * (Left to the step, the samples behind a zero crossing were stepped one after the other: their gates had been
       * evaluated against the phase of the symbol before.) */
      while ((key == NFC_FK_A_BPSK_SYMBOL || key == NFC_FK_B_SYMBOL) && from + run < n)
      {
         const uint32_t g = from + run;
         const uint32_t clk = clock0 + 1u + g;
         const NfcDecodeRegs &d = s.u.decode;
         const float phase = lds->sum[0][g];
         const float lastPhase = d.lock.lastPhase;

         const bool cross = !d.lock.auxTime && ((phase > 0.0f && lastPhase < 0.0f) || (phase < 0.0f && lastPhase > 0.0f));

         if (!cross || clk == d.lock.sync)
            break; /* the synchronisation sample: the step's */

         const uint32_t p2 = d.rt.p2;

         NFC_WAVE_READ_FENCE(); /* (the record is about to change) */
         NFC_WAVE_COUNT(49u, 0u, 1u);

         NFC_WAVE_UNIFORM_BEGIN
         {
            lds->u.s.u.decode.lock.auxTime = clk;
            lds->u.s.u.decode.lock.sync = clk + p2;
            lds->u.s.u.decode.lock.lastPhase = phase;
         }
         NFC_WAVE_UNIFORM_END

         const bool again = nfc_wave_locked_gate(c, lds, clock0, key);
         const uint64_t rest = NFC_WAVE_BALLOT(again && lane > g && lane < n);

         gated = rest >> from;
         run = rest ? (uint32_t)__builtin_ctzll(rest) - from : n - from;
      }

      /* The start of an NFC-A BPSK answer (nfca_listen_bpsk_start): while the integrated phase stays above the threshold -
       * the three or four bit periods of the preamble - every sample notes where the burst began (once) and pushes the end of
       * its window half a symbol on, nothing else; the window cannot end, nor the phase turn negative, on such a sample. A run
       * of them leaves what its last one leaves. */
      while (key == NFC_FK_A_BPSK_START && from + run < n)
      {
         const uint32_t g = from + run;
         const NfcDecodeRegs &d = s.u.decode;
         const float phase = lds->sum[0][lane];
         const float deepAt = lds->ring[NFC_R_DEPTH + (t & NFC_FMASK)];

         const bool plain = t > d.guardEnd && t <= d.waitingEnd && !(deepAt > c.minDepth[0]) && phase > d.lock.thr && !(phase < 0.0f);
         const uint64_t ok = NFC_WAVE_BALLOT(plain && lane >= g && lane < n) >> g;
         const uint32_t many = (ok & 1ull) ? (~ok ? (uint32_t)__builtin_ctzll(~ok) : 64u) : 0u; /* samples g .. g + many - 1 */

         if (many == 0u)
            break;

         const uint32_t first = clock0 + 1u + g, last = clock0 + g + many;
         const uint32_t p2 = d.rt.p2;

         NFC_WAVE_READ_FENCE();
         NFC_WAVE_COUNT(49u, 0u, many);

         NFC_WAVE_UNIFORM_BEGIN
         {
            if (!lds->u.s.u.decode.lock.symStart)
               lds->u.s.u.decode.lock.symStart = first;
            lds->u.s.u.decode.lock.winEnd = last + p2;
         }
         NFC_WAVE_UNIFORM_END

         const bool again = nfc_wave_locked_gate(c, lds, clock0, key);
         const uint64_t rest = NFC_WAVE_BALLOT(again && lane >= g + many && lane < n);

         gated = rest >> from;
         run = rest ? (uint32_t)__builtin_ctzll(rest) - from : n - from;
      }

      /* NFC-V poll frames (nfcv_poll_symbol): a pulse correlation above the threshold that is the largest so far moves the
       * record's maximum and, with it, the end of the window; what the window holds is decided at its end, by the step */
      while (key == NFC_FK_V_POLL && from + run < n)
      {
         const uint32_t g = from + run;
         const uint32_t clk = clock0 + 1u + g;
         const NfcDecodeRegs &d = s.u.decode;

         if (clk < d.lock.winStart || clk == d.lock.winEnd)
            break;

         float g0, g1;
         nfc_wave_s0s1(c, lds, key, 0u, g, g0, g1);
         const float q = g0 / (float)d.rt.p2; /* (c2 - sum) / p2: nfcv_pulse_apply */

         if (!(q > d.lock.thr && q > d.lock.peak))
            break; /* (nothing to do here after all: the step finds that out as well) */

         const uint32_t p4 = d.rt.p4;

         NFC_WAVE_READ_FENCE();
         NFC_WAVE_COUNT(49u, 0u, 1u);

         NFC_WAVE_UNIFORM_BEGIN
         {
            lds->u.s.u.decode.lock.peak = q;
            lds->u.s.u.decode.lock.peakTime = clk;
            lds->u.s.u.decode.lock.winEnd = clk + p4;
         }
         NFC_WAVE_UNIFORM_END

         const bool again = nfc_wave_locked_gate(c, lds, clock0, key);
         const uint64_t rest = NFC_WAVE_BALLOT(again && lane > g && lane < n);

         gated = rest >> from;
         run = rest ? (uint32_t)__builtin_ctzll(rest) - from : n - from;
      }

      /* the start of an NFC-V answer (nfcv_listen_start): a burst correlation beyond the threshold on either side that is
       * the largest so far moves the record's extreme and the end of its window; the window's end is the step's */
      while (key == NFC_FK_V_START && from + run < n)
      {
         const uint32_t g = from + run;
         const uint32_t clk = clock0 + 1u + g;
         const NfcDecodeRegs &d = s.u.decode;
         const float deepAt = lds->ring[NFC_R_DEPTH + (clk & NFC_FMASK)];

         if (!(clk > d.guardEnd && clk <= d.waitingEnd && !(deepAt > c.maxDepth[3]) && clk >= d.lock.winStart))
            break;

         float g0, g1;
         nfc_wave_s0s1(c, lds, key, 0u, g, g0, g1);

         float peak = d.lock.peak;
         bool moved = false;

         if (g0 < -d.lock.thr && g0 < peak)
         {
            peak = g0;
            moved = true;
         }
         if (g0 > d.lock.thr && g0 > peak)
         {
            peak = g0;
            moved = true;
         }

         if (!moved)
            break; /* (the end of the window, or nothing at all: the step's) */

         const uint32_t p8 = d.rt.p8;

         NFC_WAVE_READ_FENCE();
         NFC_WAVE_COUNT(49u, 0u, 1u);

         NFC_WAVE_UNIFORM_BEGIN
         {
            lds->u.s.u.decode.lock.peak = peak;
            lds->u.s.u.decode.lock.peakTime = clk;
            lds->u.s.u.decode.lock.winEnd = clk + p8;
         }
         NFC_WAVE_UNIFORM_END

         const bool again = nfc_wave_locked_gate(c, lds, clock0, key);
         const uint64_t rest = NFC_WAVE_BALLOT(again && lane > g && lane < n);

         gated = rest >> from;
         run = rest ? (uint32_t)__builtin_ctzll(rest) - from : n - from;
      }

      /* the same for the pulse tracker of an NFC-A 106 k listen-frame start (nfca_listen_ask_track): inside the waiting time,
       * past the sample that seeds the threshold, modulation not deeper than a card's */
      while (key == NFC_FK_A_ASK_START && from + run < n)
      {
         const uint32_t g = from + run;
         const uint32_t clk = clock0 + 1u + g;
         const NfcDecodeRegs &d = s.u.decode;
         const float deepAt = lds->ring[NFC_R_DEPTH + (clk & NFC_FMASK)];

         if (!(clk > d.guardEnd && clk <= d.waitingEnd && !(deepAt > c.minDepth[0])))
            break;

         float g0, g1;
         nfc_wave_s0s1(c, lds, key, 0u, g, g0, g1);

         NfcMod m = *(const NfcMod *)&lds->u.s.u.decode.lock;
         const NfcRate rt = d.rt;

         if (nfca_listen_ask_track(m, rt, clk, g0))
            break; /* start of frame: the step's */

         NFC_WAVE_READ_FENCE(); /* (the record is about to change) */
         NFC_WAVE_COUNT(49u, 0u, 1u);

         NFC_WAVE_UNIFORM_BEGIN
         {
            m.acc = lds->u.s.u.decode.lock.acc; /* (the running sum is the commit's) */
            *(NfcMod *)&lds->u.s.u.decode.lock = m;
         }
         NFC_WAVE_UNIFORM_END

         const bool again = nfc_wave_locked_gate(c, lds, clock0, key);
         const uint64_t rest = NFC_WAVE_BALLOT(again && lane > g && lane < n);

         gated = rest >> from;
         run = rest ? (uint32_t)__builtin_ctzll(rest) - from : n - from;
      }

      while (key == NFC_FK_F_START && from + run < n)
      {
         const uint32_t g = from + run;
         const uint32_t clk = clock0 + 1u + g;
         const NfcDecodeRegs &d = s.u.decode;

         /* (nfcf_listen_start's own comparisons: past the guard time, the threshold seeded; not timed out; window open) */
         if (!(clk > d.guardEnd && clk <= d.waitingEnd && clk >= d.lock.winStart))
            break;

         float g0, g1;
         nfc_wave_s0s1(c, lds, key, 0u, g, g0, g1);
         const float sd = nfc_abs(g0 - g1) / (float)d.rt.p2;

         NfcMod m = *(const NfcMod *)&lds->u.s.u.decode.lock;
         NfcStreamState at;
         at.clock = clk;
         uint32_t polarity = 0;
         const NfcRate rt = d.rt;

         if (nfcf_track_preamble(at, m, rt, sd, g0, sd >= m.thr, polarity, nullptr))
            break; /* start of frame: the step's */

         NFC_WAVE_READ_FENCE(); /* (the record is about to change) */
         NFC_WAVE_COUNT(49u, 0u, 1u);

         NFC_WAVE_UNIFORM_BEGIN
         {
            m.acc = lds->u.s.u.decode.lock.acc; /* (the running sum is the commit's) */
            *(NfcMod *)&lds->u.s.u.decode.lock = m;
         }
         NFC_WAVE_UNIFORM_END

         /* the gates of the samples behind it, for the record as it now stands */
         const bool again = nfc_wave_locked_gate(c, lds, clock0, key);
         const uint64_t rest = NFC_WAVE_BALLOT(again && lane > g && lane < n);

         gated = rest >> from;
         run = rest ? (uint32_t)__builtin_ctzll(rest) - from : n - from;
      }
   }

   /* kept for the caller: the sample after a stepped one is stepped too when it was gated here */
   NFC_WAVE_UNIFORM_BEGIN
   {
      lds->u.gatedLo = (uint32_t)gated;
      lds->u.gatedHi = (uint32_t)(gated >> 32);
      lds->u.gatedFrom = from;
   }
   NFC_WAVE_UNIFORM_END

   /* the sample after the run is one to step: the caller goes there straight from here (false) */
   const bool more = from + run < n;

   if (more)
   {
      if (key == NFC_FK_SEARCH && (which & 0xFFu) == 0u)
         NFC_WAVE_COUNT(47u, 0u, 1u); /* search: unarmed sample or carrier frame */
      /* the search step only has to ask the detectors whose gates are up (the others take their early exits) */
      NFC_WAVE_UNIFORM_BEGIN
      {
         lds->u.which = key == NFC_FK_SEARCH ? (which & 0xFFFu) : 0xFFFFFFFFu; /* (bits 8 .. 11: what an NFC-F detector that is not asked still leaves) */
         lds->u.whichAt = from + run;
      }
      NFC_WAVE_UNIFORM_END
#ifdef NFC_WAVE_COUNT_DETECTORS
      if (lane == from)
         for (uint32_t b = 0; b < 8u; b++)
            if ((which >> b) & 1u)
               NFC_WAVE_COUNT_DETECTORS(b);
#endif
   }

   if (run == 0u)
      return false;

   NFC_WAVE_COUNT(key, 0u, run);

   /* ---- commit: ring entries by the lanes of the run, then the state ---- */
   NFC_WAVE_TICK(lds, 4u);
   const uint32_t last = from + run - 1u;
   const uint32_t never = s.clock - 0x40000000u;

   NFC_WAVE_BARRIER();

   if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
   {
      const bool all = key == NFC_FK_UPKEEP; /* the warm-up keeps every correlator up (nfc_step_upkeep) */

      if (all || (c.enabled & 1u))
      {
         nfc_wave_ring_commit(lds, 0u, clock0, from, run, c.a[0].p1, c.corrOffset[0], s.posA[0], never);
         nfc_wave_ring_commit(lds, 1u, clock0, from, run, c.a[1].p1, c.corrOffset[1], s.posA[1], never);
         nfc_wave_ring_commit(lds, 2u, clock0, from, run, c.a[2].p1, c.corrOffset[2], s.posA[2], never);
      }

      if (all || (c.enabled & 4u))
      {
         nfc_wave_ring_commit(lds, 3u, clock0, from, run, c.f[1].p1, c.corrOffset[3], s.posF[0], never);
         nfc_wave_ring_commit(lds, 4u, clock0, from, run, c.f[2].p1, c.corrOffset[4], s.posF[1], never);
      }

      if (all || (c.enabled & 8u))
         nfc_wave_ring_commit(lds, 5u, clock0, from, run, c.v.p1, c.corrOffset[5], s.posV1, never);
   }
   else if (key == NFC_FK_A_POLL || key == NFC_FK_F_DATA || key == NFC_FK_F_START || key == NFC_FK_A_ASK_START || key == NFC_FK_A_ASK_SYMBOL)
      nfc_wave_ring_commit(lds, 0u, clock0, from, run, s.u.decode.rt.p1, s.u.decode.lockBase, s.u.decode.lockPos,
                           key == NFC_FK_F_START ? s.u.decode.guardEnd - s.u.decode.rt.p1 : never);
   else if (key == NFC_FK_V_POLL)
      nfc_wave_ring_commit(lds, 0u, clock0, from, run, s.u.decode.rt.p1, s.u.decode.lockBase, s.posV1, never);
   else if (key == NFC_FK_V_START || key == NFC_FK_V_SYMBOL)
      nfc_wave_ring_commit(lds, 0u, clock0, from, run, s.u.decode.rt.p0, s.u.decode.lockBase, s.posV0, never);

   const bool gathers = key == NFC_FK_A_POLL || key == NFC_FK_A_ASK_SYMBOL || key == NFC_FK_F_DATA || key == NFC_FK_V_SYMBOL;

   /* NFC-F detectors asked to reset (a clear record) somewhere in the run: the mark nfcf_detect_rate leaves */
   uint32_t marks = 0;
   if (key == NFC_FK_SEARCH)
      marks = nfc_wave_f_marks(lds, bits, marksFrom, last + 1u);

   if (gathers)
      nfc_wave_fold(c, lds, clock0, key, from, run);

   NFC_WAVE_BARRIER();

   NFC_WAVE_UNIFORM_BEGIN
   {
      NFC_WAVE_LDS NfcStreamState &w = lds->u.s;
      const uint32_t firstClock = w.clock + 1u;

      lds->flags |= marks;

      if (key == NFC_FK_SEARCH || key == NFC_FK_UPKEEP)
      {
         const bool all = key == NFC_FK_UPKEEP;

         if (all || (c.enabled & 1u))
         {
            w.u.search.detA[0].acc = lds->sum[0][last];
            w.u.search.detA[1].acc = lds->sum[1][last];
            w.u.search.detA[2].acc = lds->sum[2][last];
         }
         if (all || (c.enabled & 4u))
         {
            w.u.search.detF[0].acc = lds->sum[3][last];
            w.u.search.detF[1].acc = lds->sum[4][last];
         }
         if (all || (c.enabled & 8u))
            w.u.search.detV.acc = lds->sum[5][last];

         /* the bank has stepped on every sample of the run (nfc_search_detect / nfc_step_upkeep) */
         if (w.bankClock != firstClock - 1u)
            lds->cold.bankRun = firstClock;
         w.bankClock = w.clock + run;
      }
      else if (key == NFC_FK_A_BPSK_START || key == NFC_FK_A_BPSK_SYMBOL || key == NFC_FK_B_START || key == NFC_FK_B_SYMBOL)
         w.u.decode.lock.phaseAcc = lds->sum[0][last];
      else if (key != NFC_FK_UNARMED && key != NFC_FK_B_POLL)
         w.u.decode.lock.acc = lds->sum[0][last];

      if (gathers)
      {
         if (lds->u.pass[8] != 0.0f)
         {
            const uint32_t when = __builtin_bit_cast(uint32_t, (float)lds->u.pass[10]);
            w.u.decode.lock.peak = lds->u.pass[9];

            if (key == NFC_FK_V_SYMBOL)
            {
               /* nfcv_listen_symbol keeps the correlation's sign and the sample */
               w.u.decode.lock.c0 = lds->u.pass[11];
               w.u.decode.lock.c1 = -lds->u.pass[11];
               w.u.decode.lock.symEnd = when;
            }
            else
               w.u.decode.lock.peakTime = when;
         }

         if (lds->u.pass[12] != 0.0f && key != NFC_FK_V_SYMBOL)
         {
            if (key != NFC_FK_F_DATA)
               w.u.decode.lock.cD = lds->u.pass[13];
            w.u.decode.lock.c0 = lds->u.pass[14];
            w.u.decode.lock.c1 = lds->u.pass[15];
         }
      }

      if (w.lockTech)
         w.u.decode.lockPos = nfc_wave_wrap3(w.u.decode.lockPos + run, w.u.decode.rt.p1);

      nfc_wave_advance(c, w, run);
      w.clock += run;
      w.env = lds->env[last];
      w.avg = lds->avg[last];
      w.mdev = lds->ring[NFC_R_MDEV + (w.clock & NFC_FMASK)];

      lds->u.at = from + run;
   }
   NFC_WAVE_UNIFORM_END

   NFC_WAVE_TICK(lds, 11u);
   return !more;
}

#endif
