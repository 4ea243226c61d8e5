/*
 * nfc_tech_b.hpp — ISO14443-B / NFC-B: NRZ-L ASK (10 %) poll frames, BPSK listen frames.
 *
 * Reference behaviour being matched: src/nfc-lib/lib-lab/lab-radio/src/main/cpp/tech/NfcB.cpp
 *   detectModulation 238-432 (edge detector on the DC-removed signal, rates 106k/212k),
 *   decodePollFrame 453-567, decodeListenFrame 572-679, decodePollFrameSymbolAsk 684-762,
 *   decodeListenFrameStartBpsk 767-949, decodeListenFrameSymbolBpsk 954-1040,
 *   resetModulation 1045-1069, process* 1074-1267, checkCrc 1272-1283.
 * Included by nfc_core.hpp (device code).
 */
#ifndef NFC_AMD_TECH_B_HPP
#define NFC_AMD_TECH_B_HPP

NFC_DEV void nfcb_protocol_defaults(const NfcConfig &c, NfcTiming &t)
{
   t.maxFrameSize = 256;
   t.protoGuardTime = nfc_tu(c, 1024);            /* NFCB_FGT_DEF = TR0min */
   t.protoWaitingTime = nfc_tu(c, 256 * 16 * 16); /* NFCB_FWT_DEF */
}

NFC_DEV void nfcb_reset(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem)
{
   nfc_leave_lock(s, NFC_TECH_B);
}

/* ISO/IEC 13239 CRC_B */
NFC_DEV bool nfcb_crc_ok(const uint8_t *data, uint32_t len)
{
   if (len < 3)
      return false;

   uint32_t crc = (~nfc_crc16(data, len - 2, 0xFFFFu, true)) & 0xFFFFu;
   uint32_t res = (uint32_t)data[len - 2] | ((uint32_t)data[len - 1] << 8);
   return res == crc;
}

NFC_DEV void nfcb_process(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, uint32_t type, const uint8_t *data, uint32_t len,
                          uint32_t &flags, uint32_t &phase)
{
   static const uint16_t fsd[16] = {16, 24, 32, 40, 48, 64, 96, 128, 256, 512, 1024, 2048, 4096, 0, 0, 0};

   NfcTiming &t = mem.cold->tim[1];
   const bool poll = (type == NFC_FRAME_POLL);
   const uint32_t b0 = nfc_byte(data, len, 0);
   const bool crcOk = nfcb_crc_ok(data, len);

   t.guardTime = t.protoGuardTime;
   if (poll)
   {
      t.waitingTime = t.protoWaitingTime;
      nfc_wait_from_proto(mem, 1u);
   }

   /* REQB / WUPB and its ATQB */
   if (poll && b0 == 0x05 && len == 5)
   {
      t.lastCommand = b0, nfc_command_written(mem, 1u);
      t.maxFrameSize = 256;
      t.protoGuardTime = nfc_tu(c, 1024);
      t.protoWaitingTime = nfc_tu(c, 256 * 16 * 16);
      nfc_wait_proto_written(mem, 1u);
      t.guardTime = nfc_tu(c, 1024);   /* NFCB_TR0_MIN  */
      t.waitingTime = nfc_tu(c, 7680); /* NFCB_FWT_ATQB */
      nfc_wait_overridden(mem, 1u);
      phase = NFC_PHASE_SELECTION;
      if (!crcOk)
         flags |= NFC_FLAG_CRC;
   }
   else if (!poll && t.lastCommand == 0x05)
   {
      uint32_t fsdi = (nfc_byte(data, len, 10) >> 4) & 0x0f;
      uint32_t fwi = (nfc_byte(data, len, 11) >> 4) & 0x0f;

      t.maxFrameSize = fsd[fsdi];
      t.protoWaitingTime = nfc_tu(c, 4096 << fwi);
      nfc_wait_proto_written(mem, 1u);

      phase = NFC_PHASE_SELECTION;
      if (!crcOk)
         flags |= NFC_FLAG_CRC;
   }
   /* ATTRIB */
   else if (poll && b0 == 0x1d && len > 10)
   {
      static const uint16_t tr0min[4] = {0, 48 * 16, 16 * 16, 0};

      t.lastCommand = b0, nfc_command_written(mem, 1u);

      uint32_t param1 = nfc_byte(data, len, 5);
      uint32_t param2 = nfc_byte(data, len, 6);
      uint32_t tr0i = (param1 >> 6) & 0x3;
      uint32_t fsdi = param2 & 0xf;

      t.maxFrameSize = fsd[fsdi];

      if (!tr0i)
         t.protoGuardTime = nfc_tu(c, 1024);
      else
         t.protoGuardTime = nfc_tu(c, tr0min[tr0i]);

      t.waitingTime = nfc_tu(c, 71680); /* NFC_FWT_ACTIVATION */
      nfc_wait_overridden(mem, 1u);

      phase = NFC_PHASE_SELECTION;
      if (!crcOk)
         flags |= NFC_FLAG_CRC;
   }
   else if (!poll && t.lastCommand == 0x1d)
   {
      phase = NFC_PHASE_SELECTION;
   }
   else
   {
      phase = NFC_PHASE_APPLICATION;
      if (!crcOk)
         flags |= NFC_FLAG_CRC;
   }

   /* chained flags are always zero for NFC-B */

   const bool locked = (s.lockTech == NFC_TECH_B);
   const uint32_t delay = locked ? s.u.decode.rt.delay : 0u;

   if (poll)
   {
      if (locked)
      {
         s.u.decode.guardEnd = s.u.decode.frameEnd + t.guardTime + delay;
         s.u.decode.waitingEnd = s.u.decode.frameEnd + t.waitingTime + delay;
         s.u.decode.frameType = NFC_FRAME_LISTEN;
         s.u.decode.maxFrame = t.maxFrameSize;
      }
   }
   else
   {
      if (locked)
         s.u.decode.guardEnd = s.u.decode.frameEnd + t.guardTime + delay;

      s.u.decode.frameType = 0;
      t.lastCommand = 0, nfc_command_written(mem, 1u);
   }

   s.u.decode.frameStart = 0;
   s.u.decode.frameEnd = 0;
}

/* history reads of the edge detectors (rate 0 looks at the current sample, rate 1 one 106k symbol back) */
struct NfcTapsB
{
   float edge[2];
   float deep[2];
};

NFC_DEV void nfcb_load_taps(const NfcConfig &c, const NfcStreamState &s, const NfcLaneMem &mem, NfcTapsB &taps)
{
   const uint32_t slot0 = (s.clock - c.b[0].delay) & NFC_FMASK;
   const uint32_t slot1 = (s.clock - c.b[1].delay) & NFC_FMASK;
   taps.edge[0] = NFC_AT(mem, NFC_R_FILT, slot0);
   taps.deep[0] = NFC_AT(mem, NFC_R_DEPTH, slot0);
   taps.edge[1] = NFC_AT(mem, NFC_R_FILT, slot1);
   taps.deep[1] = NFC_AT(mem, NFC_R_DEPTH, slot1);
}

/* ---- search: SOF = falling edge, 10-11 etu low, rising edge, 2-3 etu high, falling edge ----
 * returns 0 = keep searching, 1 = locked, 2 = abandon this sample for the remaining rates */
template <int R>
NFC_DEV int nfcb_detect_decide(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float edge, float deep);

template <int R>
NFC_DEV int nfcb_detect_rate(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcTapsB &taps, const NfcNow &now)
{
   const NfcRate &rt = c.b[R];

   /* with no delay the sample of interest is the one the front end has just produced */
   return nfcb_detect_decide<R>(c, s, mem, rt.delay ? taps.edge[R] : now.filt, rt.delay ? taps.deep[R] : now.depth);
}

/* The start-of-frame tracker of one rate on its record alone. clock / env: the decoder's at this sample; edge / deep:
 * DC-removed signal and modulation depth at the detector's decode point. Returns 0 = keep searching, 1 = start of frame
 * recognised (the record still holds it: the caller locks), 2 = abandon this sample for the remaining rates.
 * Only a record past its second edge (symEnd set) can return anything but 0. */
template <int R>
NFC_DEV int nfcb_track(const NfcConfig &c, NfcDetB &m, uint32_t clock, float env, float edge, float deep)
{
   const NfcRate &rt = c.b[R];

   /* one branch for the common case: no start of frame is being tracked, no reset, no falling edge beyond the threshold
    * that is a new extreme, and the (closed) window does not end now. The threshold of that state is recomputed on every
    * sample before it is used, so not storing it on the early exit changes nothing. */
   const bool reset = deep > c.maxDepth[1] || (m.auxTime && clock > m.auxTime + rt.p1);

   /* (a reset of a record that is clear already - every sample of a 100 % ASK pause asks for one - changes nothing) */
   const bool clear = (m.symStart | m.symEnd | m.winStart | m.winEnd | m.auxTime | nfc_bits(m.aux)) == 0u;

   if ((!reset || clear) && !m.symStart && !(edge < -(env * c.minDepth[1]) && edge < m.aux) && clock != m.winEnd)
      return 0;
   if (reset)
   {
      m.symStart = 0; m.symEnd = 0; m.winStart = 0; m.winEnd = 0;
      m.auxTime = 0; m.aux = 0;
   }

   if (!m.symStart)
   {
      m.thr = env * c.minDepth[1];

      if (edge < -m.thr && edge < m.aux)
      {
         m.aux = edge;
         m.auxTime = clock;
         m.winEnd = clock + rt.p4;
      }

      if (clock != m.winEnd)
         return 0;

      m.symStart = m.auxTime - rt.p8;
      m.winStart = m.symStart + (10 * rt.p1) - rt.p2;
      m.winEnd = m.symStart + (11 * rt.p1) + rt.p2;
      m.thr = nfc_abs(m.aux * 0.5f);
      m.aux = 0;
      m.auxTime = 0;
      return 0;
   }

   if (!m.symEnd)
   {
      if (clock < m.winStart)
      {
         if (edge > m.thr)
         {
            m.symStart = 0; m.symEnd = 0; m.winStart = 0; m.winEnd = 0;
            m.auxTime = 0; m.aux = 0;
         }
         return 0;
      }

      if (edge > m.thr && edge > m.aux)
      {
         m.aux = edge;
         m.auxTime = clock;
         m.winEnd = clock + rt.p4;
      }

      if (clock != m.winEnd)
         return 0;

      if (!m.auxTime)
      {
         m.symStart = 0; m.symEnd = 0; m.winStart = 0; m.winEnd = 0; m.aux = 0;
         return 0;
      }

      m.symEnd = m.auxTime;
      m.winStart = m.auxTime + (2 * rt.p1) - rt.p2;
      m.winEnd = m.auxTime + (3 * rt.p1) + rt.p2;
      m.thr = nfc_abs(m.aux) / 2;
      m.aux = 0;
      m.auxTime = 0;
      return 0;
   }

   if (clock < m.winStart)
   {
      if (edge < -m.thr)
      {
         m.symStart = 0; m.symEnd = 0; m.winStart = 0; m.winEnd = 0;
         m.auxTime = 0; m.aux = 0;
      }
      return 0;
   }

   if (edge < -m.thr && m.aux > edge)
   {
      m.aux = edge;
      m.auxTime = clock;
      m.winEnd = clock + rt.p4;
   }

   if (clock != m.winEnd)
      return 0;

   if (!m.auxTime)
   {
      m.symStart = 0; m.symEnd = 0; m.winStart = 0; m.winEnd = 0;
      m.auxTime = 0; m.aux = 0;
      return 2; /* the reference leaves the rate loop here (NfcB.cpp:396) */
   }

   return 1;
}

/* edge / deep: DC-removed signal and modulation depth at the detector's decode point */
template <int R>
NFC_DEV int nfcb_detect_decide(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float edge, float deep)
{
   const NfcRate &rt = c.b[R];
   NfcDetB &m = s.u.search.detB[R];

   const int tracked = nfcb_track<R>(c, m, s.clock, s.env, edge, deep);

   if (tracked != 1)
      return tracked;

   /* SOF recognised: lock; the first bit is sampled half a symbol after the last edge, no search window yet */
   const uint32_t symStart = m.symStart, symEnd = m.auxTime;
   const float aux = m.aux;

   NfcDecodeRegs &out = nfc_take_lock(mem, rt, (uint32_t)R, 0, 0);
   NfcMod &d = out.lock;
   d.symStart = symStart;
   d.symEnd = symEnd;
   d.sync = symEnd + rt.p2;
   d.thr = nfc_abs(aux * 0.5f);

   out.frameType = NFC_FRAME_POLL;
   out.frameRate = rt.symbolsPerSecond;
   out.frameStart = symStart - rt.delay;
   out.frameEnd = 0;

   return 1;
}

/* the caller has checked that the search bank is armed (nfc_search_detect) */
NFC_DEV bool nfcb_detect(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcTapsB &taps, const NfcNow &now)
{
   int r0 = nfcb_detect_rate<0>(c, s, mem, taps, now);
   if (r0)
      return r0 == 1;

   return nfcb_detect_rate<1>(c, s, mem, taps, now) == 1;
}

/* ---- poll symbols: sample modulation depth at bit centres, resync on edges ---- */
NFC_DEV uint32_t nfcb_poll_symbol(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   float edge = taps.f0;
   float deep = taps.d0;

   if (s.clock > m.winStart && s.clock < m.winEnd)
   {
      edge = nfc_abs(edge);

      if (edge > m.thr && m.aux < edge)
      {
         m.aux = edge;
         m.sync = s.clock + rt.p2;
      }
   }

   if (s.clock != m.sync)
      return SYM_NONE;

   m.symStart = m.symEnd;
   m.symEnd = m.sync + rt.p2;
   m.winStart = m.sync + rt.p4;
   m.winEnd = m.winStart + rt.p2;
   m.sync = m.sync + rt.p1;
   m.aux = 0;

   if (deep > c.minDepth[1])
   {
      s.u.decode.symValue = 0;
      s.u.decode.symPattern = B_L;
   }
   else
   {
      s.u.decode.symValue = 1;
      s.u.decode.symPattern = B_H;
   }

   s.u.decode.symStart = m.symStart - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.delay;

   return s.u.decode.symPattern;
}

/* ---- listen SOF: TR1 subcarrier, then two phase changes (S1, S2) ---- */
NFC_DEV uint32_t nfcb_listen_start(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   const float deep = now.depth;
   const float guardDev = taps.m0;
   const NfcPhase p = nfc_phase_product(mem, s.clock, rt, taps.f0, taps.f1, taps.pp);

   nfc_phase_integrate(m, p);

   if (s.clock < s.u.decode.guardEnd)
      return SYM_NONE;

   if (s.clock == s.u.decode.guardEnd)
      m.thr = guardDev;

   if (s.clock > s.u.decode.waitingEnd)
      return SYM_TIMEOUT;

   if (deep > c.maxDepth[1])
      return SYM_TIMEOUT;

   if (s.clock < m.winStart)
      return SYM_NONE;

   if (m.phaseAcc > m.thr)
   {
      if (!m.symStart)
         m.symStart = s.clock;

      m.winEnd = s.clock + rt.p2;
   }

   if (s.clock != m.winEnd && m.phaseAcc > 0)
      return SYM_NONE;

   /* durations are compared unsigned against the protocol limits (int vs unsigned int in the reference) */
   const uint32_t tr1Min = nfc_tu(c, 1024), tr1Max = nfc_tu(c, 3200);
   const uint32_t s1Min = nfc_tu(c, 1272), s1Max = nfc_tu(c, 1416);
   const uint32_t s2Min = nfc_tu(c, 248), s2Max = nfc_tu(c, 392);

   if (m.stage == 0)
   {
      uint32_t length = s.clock - m.symStart;

      if (length < tr1Min || length > tr1Max)
      {
         m.stage = 0; m.winStart = 0; m.winEnd = 0; m.symStart = 0; m.symEnd = 0;
         return SYM_NONE;
      }

      m.symEnd = s.clock;
      m.stage = 1;
      m.winStart = s.clock + rt.p1 + rt.p4;
      m.winEnd = 0;
      return SYM_NONE;
   }

   if (m.stage == 1)
   {
      uint32_t length = s.clock - m.symEnd;

      if (length < s1Min || length > s1Max)
      {
         m.stage = 0; m.winStart = 0; m.winEnd = 0; m.symStart = 0; m.symEnd = 0;
         return SYM_NONE;
      }

      m.symEnd = s.clock;
      m.stage = 2;
      m.winStart = s.clock + rt.p1 + rt.p4;
      m.winEnd = 0;
      return SYM_NONE;
   }

   if (m.stage == 2)
   {
      uint32_t length = s.clock - m.symEnd;

      if (length < s2Min || length > s2Max)
      {
         m.stage = 0; m.winStart = 0; m.winEnd = 0; m.symStart = 0; m.symEnd = 0;
         return SYM_NONE;
      }

      m.symEnd = s.clock;
      m.sync = s.clock + rt.p2;
      m.lastPhase = m.phaseAcc;
      m.phaseThr = nfc_abs(m.aux * 0.25f);
      m.winStart = 0;
      m.winEnd = 0;
      m.aux = 0;

      s.u.decode.symValue = 1;
      s.u.decode.symStart = m.symStart - rt.p1 - rt.delay;
      s.u.decode.symEnd = m.symEnd - rt.p1 - rt.delay;
      s.u.decode.symPattern = B_S;

      return B_S;
   }

   /* any other stage value: the reference's switch has no case and simply keeps consuming samples */
   return SYM_NONE;
}

/* ---- listen symbols: BPSK phase at bit centres ---- */
NFC_DEV uint32_t nfcb_listen_symbol(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   const NfcPhase p = nfc_phase_product(mem, s.clock, rt, taps.f0, taps.f1, taps.pp);
   nfc_phase_integrate(m, p);

   if (!m.auxTime)
   {
      if ((m.phaseAcc > 0 && m.lastPhase < 0) || (m.phaseAcc < 0 && m.lastPhase > 0))
      {
         m.auxTime = s.clock;
         m.sync = s.clock + rt.p2;
         m.lastPhase = m.phaseAcc;
      }
   }

   if (s.clock != m.sync)
      return SYM_NONE;

   if (nfc_abs(m.phaseAcc) < nfc_abs(m.phaseThr))
      return B_O;

   m.symStart = m.symEnd;
   m.symEnd = m.sync + rt.p2;
   m.sync = m.sync + rt.p1;
   m.lastPhase = m.phaseAcc;
   m.auxTime = 0;

   if (m.phaseAcc < -m.phaseThr)
   {
      s.u.decode.symValue = !s.u.decode.symValue;
      s.u.decode.symPattern = (s.u.decode.symPattern == B_M) ? B_N : B_M;
   }
   else
   {
      m.phaseThr = m.phaseAcc * 0.25f;
   }

   s.u.decode.symStart = m.symStart - rt.p1 - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.p1 - rt.delay;

   return s.u.decode.symPattern;
}

/* ---- one sample in locked NFC-B mode ---- */
NFC_DEV void nfcb_decode(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now, const NfcDecTaps &taps)
{

   if (s.u.decode.frameType == NFC_FRAME_POLL)
   {
      uint32_t pattern = nfcb_poll_symbol(c, s, mem, taps);

      if (pattern <= SYM_TIMEOUT)
         return;

      bool frameEnd = false, truncated = false, streamError = false;

      if (s.u.decode.bsBits == 9 && !s.u.decode.bsData && pattern == B_L)
         frameEnd = true;
      else if (s.u.decode.bsBits == 9 && pattern == B_L)
         streamError = true;
      else if (s.u.decode.bsBits == 0 && pattern == B_H && s.u.decode.bsSkip == 6)
         streamError = true;
      else if (s.u.decode.bsBytes == s.u.decode.maxFrame)
         truncated = true;
      else if ((s.u.decode.bsBits == 0 && pattern == B_H) && ++s.u.decode.bsSkip)
         return; /* extra guard time between characters */

      if (frameEnd || streamError || truncated)
      {
         if (s.u.decode.bsBytes > 2)
         {
            s.u.decode.frameEnd = s.u.decode.symEnd;

            nfc_pend_frame(s, NFC_FRAME_POLL, (truncated || streamError) ? NFC_FLAG_TRUNCATED : 0);
            return;
         }

         nfcb_reset(c, s, mem);
         return;
      }

      if (s.u.decode.bsBits < 9)
      {
         if (s.u.decode.bsBits > 0)
            s.u.decode.bsData |= (s.u.decode.symValue << (s.u.decode.bsBits - 1));

         s.u.decode.bsBits++;
      }
      else
      {
         nfc_push_byte(mem, s, s.u.decode.bsData);
         s.u.decode.bsData = 0;
         s.u.decode.bsBits = 0;
         s.u.decode.bsSkip = 0;
      }

      return;
   }

   if (s.u.decode.frameType != NFC_FRAME_LISTEN)
      return;

   if (!s.u.decode.frameStart)
   {
      uint32_t pattern = nfcb_listen_start(c, s, mem, now, taps);

      if (pattern == B_S || pattern == SYM_TIMEOUT)
         nfc_wait_ended(mem, s, 1u, pattern == SYM_TIMEOUT && s.clock > s.u.decode.waitingEnd);

      if (pattern == B_S)
         s.u.decode.frameStart = s.u.decode.symStart;
      else if (pattern == SYM_TIMEOUT)
         nfcb_reset(c, s, mem);

      return;
   }

   uint32_t pattern = nfcb_listen_symbol(c, s, mem, taps);

   if (pattern <= SYM_TIMEOUT)
      return;

   bool frameEnd = false, truncated = false, streamError = false;

   if (s.u.decode.bsBits == 9 && !s.u.decode.bsData && pattern == B_M)
      frameEnd = true;
   else if ((s.u.decode.bsBits == 0 && pattern == B_N) || (s.u.decode.bsBits == 9 && pattern == B_M))
      streamError = true;
   else if (s.u.decode.bsBytes == s.u.decode.maxFrame)
      truncated = true;

   if (frameEnd || streamError || truncated)
   {
      if (s.u.decode.bsBytes > 0)
      {
         s.u.decode.frameEnd = s.u.decode.symEnd + nfc_tu(c, 352); /* EOS is not tracked to its end */

         nfc_pend_frame(s, NFC_FRAME_LISTEN, (truncated || streamError) ? NFC_FLAG_TRUNCATED : 0);
         return;
      }

      nfcb_reset(c, s, mem);
      return;
   }

   if (s.u.decode.bsBits < 9)
   {
      if (s.u.decode.bsBits > 0)
         s.u.decode.bsData |= (s.u.decode.symValue << (s.u.decode.bsBits - 1));

      s.u.decode.bsBits++;
   }
   else
   {
      nfc_push_byte(mem, s, s.u.decode.bsData);
      s.u.decode.bsData = 0;
      s.u.decode.bsBits = 0;
   }
}

#endif
