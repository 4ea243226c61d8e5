/*
 * nfc_tech_v.hpp — ISO15693 / NFC-V: pulse-position (1 of 4 / 1 of 256) poll frames,
 * single-subcarrier Manchester listen frames.
 *
 * Reference behaviour being matched: src/nfc-lib/lib-lab/lab-radio/src/main/cpp/tech/NfcV.cpp
 *   configurePulse 220-234, detectModulation 236-435, decodePollFrame 450-556,
 *   decodeListenFrame 561-667, decodePollFrameSymbolPpm 672-795, decodeListenFrameStartAsk 800-980,
 *   decodeListenFrameSymbolAsk 985-1074, resetModulation 1079-1103, process 1108-1183, checkCrc 1193-1205.
 * Included by nfc_core.hpp (device code).
 */
#ifndef NFC_AMD_TECH_V_HPP
#define NFC_AMD_TECH_V_HPP

NFC_DEV void nfcv_protocol_defaults(const NfcConfig &c, NfcTiming &t)
{
   t.maxFrameSize = 256;
   t.protoGuardTime = nfc_tu(c, 1024);            /* NFCV_FGT_DEF */
   t.protoWaitingTime = nfc_tu(c, 256 * 16 * 16); /* NFCV_FWT_DEF */
}

NFC_DEV void nfcv_reset(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem)
{
   nfc_leave_lock(s, NFC_TECH_V);
}

NFC_DEV bool nfcv_crc_ok(const uint8_t *data, uint32_t len)
{
   if (len < 3)
      return false;

   uint32_t crc = (~nfc_crc16(data, len - 2, 0xFFFFu, true)) & 0xFFFFu;
   uint32_t res = (uint32_t)data[len - 2] | ((uint32_t)data[len - 1] << 8);
   return res == crc;
}

NFC_DEV void nfcv_process(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, uint32_t type, const uint8_t *data, uint32_t len,
                          uint32_t &flags, uint32_t &phase)
{
   NfcTiming &t = mem.cold->tim[3];
   const bool poll = (type == NFC_FRAME_POLL);

   t.guardTime = t.protoGuardTime;
   if (poll)
   {
      t.waitingTime = t.protoWaitingTime;
      nfc_wait_from_proto(mem, 3u);
   }

   phase = NFC_PHASE_APPLICATION;
   if (!nfcv_crc_ok(data, len))
      flags |= NFC_FLAG_CRC;

   const bool locked = (s.lockTech == NFC_TECH_V);

   if (poll)
   {
      if (locked)
      {
         /* note the sign: the poll side runs on the delayed signal (NfcV.cpp:1145-1148) */
         s.u.decode.guardEnd = s.u.decode.frameEnd + t.guardTime - c.v.delay;
         s.u.decode.waitingEnd = s.u.decode.frameEnd + t.waitingTime - c.v.delay;
         s.u.decode.frameType = NFC_FRAME_LISTEN;
         s.u.decode.maxFrame = t.maxFrameSize;
      }
   }
   else
   {
      if (locked)
         s.u.decode.guardEnd = s.u.decode.frameEnd + t.guardTime + c.v.delay;

      s.u.decode.frameType = 0;
      t.lastCommand = 0, nfc_command_written(mem, 3u);
   }

   s.u.decode.frameStart = 0;
   s.u.decode.frameEnd = 0;
}

/* history reads of the pulse correlator: box sum over p2 of the raw signal delayed by two symbols; S0 compares
 * it with half a symbol before */
struct NfcTapsV
{
   NfcTap t;
};

NFC_DEV void nfcv_load_taps(const NfcConfig &c, const NfcStreamState &s, const NfcLaneMem &mem, NfcTapsV &taps)
{
   taps.t = nfc_tap_raw(mem, s.clock, c.v, c.corrOffset[5], s.posV1, false);
}

template <class M>
NFC_DEV float nfcv_pulse_apply(const NfcLaneMem &mem, M &m, const NfcTap &t, uint32_t base, uint32_t pos, const NfcRate &rt)
{
   m.acc += t.in;
   m.acc -= t.out;

   NFC_AT(mem, NFC_R_CORR, base + pos) = m.acc;

   return (t.c2 - m.acc) / (float)rt.p2;
}

NFC_DEV bool nfcv_detect_decide(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float num, float raw);

/* the caller has checked that the search bank is armed (nfc_search_detect) */
NFC_DEV bool nfcv_detect(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcTapsV &taps, const NfcNow &now)
{
   NfcDetV &m = s.u.search.detV;

   m.acc += taps.t.in;
   m.acc -= taps.t.out;
   NFC_AT(mem, NFC_R_CORR, c.corrOffset[5] + s.posV1) = m.acc;

   return nfcv_detect_decide(c, s, mem, taps.t.c2 - m.acc, taps.t.in);
}

/* what the detector does with this sample's pulse correlation (num = ring entry half a symbol back - box sum) and the
 * delayed raw sample; the correlator itself has been stepped by the caller */
NFC_DEV bool nfcv_detect_decide(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float num, float raw)
{
   const NfcRate &rt = c.v;
   NfcDetV &m = s.u.search.detV;

   const float minimumCorrelation = s.env * c.corrThreshold[3];

   /* one branch for the common case (see nfca_detect_rate) */
   const bool timeout = m.peakTime && s.clock > m.peakTime + rt.p0;
   const bool eventful = s.clock >= m.winStart && (nfc_may_exceed(num, (float)rt.p2, minimumCorrelation) || s.clock == m.winEnd);

   if (!timeout && !eventful)
      return false;

   if (timeout)
   {
      m.symStart = 0; m.winStart = 0; m.winEnd = 0;
      m.aux = 0; m.peakTime = 0; m.peak = 0;
   }

   if (s.clock < m.winStart)
      return false;

   if (nfc_may_exceed(num, (float)rt.p2, minimumCorrelation))
   {
      const float s0 = num / (float)rt.p2;

      if (s0 > minimumCorrelation)
      {
         if (s0 > m.peak)
         {
            m.peak = s0;
            m.peakTime = s.clock;
            m.winEnd = s.clock + rt.p4;
         }

         /* modulation depth one eighth of a symbol back: only needed while a pulse is being tracked */
         const float deep = NFC_F_AT(mem, NFC_R_DEPTH, s.clock - rt.delay - rt.p8);

         if (deep > m.aux)
            m.aux = deep;
      }
   }

   if (s.clock != m.winEnd)
      return false;

   if (raw < minimumCorrelation || m.peakTime == 0 || m.aux < c.minDepth[3])
   {
      m.symStart = 0; m.winStart = 0; m.winEnd = 0;
      m.peakTime = 0; m.peak = 0; m.aux = 0;
      return false;
   }

   if (!m.symStart)
   {
      m.symStart = m.peakTime - rt.p2;
      m.winStart = m.symStart + (2 * rt.p1);
      m.winEnd = m.symStart + (4 * rt.p1);
      m.peakTime = 0; m.peak = 0; m.aux = 0;
      return false;
   }

   uint32_t symEnd, length, rate, code;

   if (m.peakTime > (m.symStart + 3 * rt.p1 - rt.p8) && m.peakTime < (m.symStart + 3 * rt.p1 + rt.p8))
   {
      symEnd = m.peakTime + rt.p1;
      length = (uint32_t)c.vLen2;
      rate = rt.symbolsPerSecond / 2;
      code = 0;
   }
   else if (m.peakTime > (m.symStart + 4 * rt.p1 - rt.p8) && m.peakTime < (m.symStart + 4 * rt.p1 + rt.p8))
   {
      symEnd = m.peakTime;
      length = (uint32_t)c.vLen8;
      rate = rt.symbolsPerSecond / 32;
      code = 1;
   }
   else
   {
      m.symStart = 0; m.winStart = 0; m.winEnd = 0;
      m.peakTime = 0; m.peak = 0; m.aux = 0;
      return false;
   }

   /* SOF (two pulses) recognised: lock (the detector record is about to be parked: take what is needed) */
   const uint32_t symStart = m.symStart, pos = s.posV1;
   const float acc = m.acc, aux = m.aux;

   NfcDecodeRegs &out = nfc_take_lock(mem, rt, 0, c.corrOffset[5], pos);
   NfcMod &d = out.lock;
   d.symStart = symStart;
   d.symEnd = symEnd;
   d.sync = symEnd;
   d.winStart = symEnd;
   d.winEnd = symEnd + length;
   d.thr = minimumCorrelation;
   d.acc = acc;
   d.aux = aux;

   out.frameRate = rate;
   out.pulseCode = code;
   out.frameType = NFC_FRAME_POLL;
   out.frameStart = symStart - rt.delay;
   out.frameEnd = 0;

   return true;
}

/* one pulse-position symbol (2 or 8 bits), NfcV.cpp:672-795 */
NFC_DEV uint32_t nfcv_poll_symbol(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   NfcTap tap;
   tap.in = taps.x0; tap.out = taps.x2; tap.c2 = taps.c2; tap.c3 = 0.0f;
   float s0 = nfcv_pulse_apply(mem, m, tap, s.u.decode.lockBase, s.posV1, rt);

   if (s.clock < m.winStart)
      return SYM_NONE;

   if (s0 > m.thr)
   {
      if (s0 > m.peak)
      {
         m.peak = s0;
         m.peakTime = s.clock;
         m.winEnd = s.clock + rt.p4;
      }
   }

   if (s.clock != m.winEnd)
      return SYM_NONE;

   /* a pulse in the first half of the second slot is the EOF marker */
   if (m.peakTime > (m.winStart + 1 * rt.p1 + rt.p4) && m.peakTime < (m.winStart + 2 * rt.p1 - rt.p4))
   {
      m.symEnd = m.peakTime + rt.p2;

      s.u.decode.symValue = 0;
      s.u.decode.symStart = m.symStart - rt.delay;
      s.u.decode.symEnd = m.symEnd - rt.delay;
      s.u.decode.symPattern = V_S;
      return V_S;
   }

   s.u.decode.symValue = 0;
   s.u.decode.symStart = m.symStart - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.delay;
   s.u.decode.symPattern = V_E;

   const int periods = s.u.decode.pulseCode ? 256 : 4;
   const int length = s.u.decode.pulseCode ? c.vLen8 : c.vLen2;
   /* the slot tables are read from the configuration in memory (dynamic index), never from a register copy */
   const int32_t *ends = s.u.decode.pulseCode ? mem.tables->vSlotEnd8 : mem.tables->vSlotEnd2;

#pragma clang loop unroll(disable)
   for (int i = 0; i < periods; i++)
   {
      const uint32_t slotEnd = (uint32_t)ends[i];

      if (m.peakTime > (m.winStart + slotEnd - rt.p4) && m.peakTime < (m.winStart + slotEnd + rt.p4))
      {
         m.symStart = m.peakTime - slotEnd;
         m.symEnd = m.symStart + (uint32_t)length;
         m.sync = m.symEnd;
         m.winStart = m.sync;
         m.winEnd = m.sync + (uint32_t)length;
         m.peakTime = 0;
         m.peak = 0;

         s.u.decode.symValue = (uint32_t)i;
         s.u.decode.symStart = m.symStart - rt.delay;
         s.u.decode.symEnd = m.symEnd - rt.delay;
         s.u.decode.symPattern = s.u.decode.pulseCode ? V_8 : V_2;
         return s.u.decode.symPattern;
      }
   }

   return V_E;
}

/* subcarrier power integrated over one symbol half (p1), ring of two symbols (p0) */
NFC_DEV float nfcv_burst_correlation(NfcStreamState &s, const NfcLaneMem &mem, NfcMod &m, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   const uint32_t cur = s.clock - rt.delay;
   const uint32_t base = s.u.decode.lockBase;
   const uint32_t pos = s.posV0;

   const float v = taps.f0;
   const float old = taps.pp;
   const float c2 = taps.c2;

   const float sq = v * v * 10.0f;

   NFC_AT(mem, NFC_R_PROD, cur & NFC_PMASK) = sq;

   m.acc += sq;
   m.acc -= old;

   NFC_AT(mem, NFC_R_CORR, base + pos) = m.acc;

   return c2 - m.acc;
}

NFC_DEV uint32_t nfcv_listen_start(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   const float deep = now.depth;
   const float guardDev = taps.m0;
   float s0 = nfcv_burst_correlation(s, mem, m, taps);

   if (s.clock < s.u.decode.guardEnd)
      return SYM_NONE;

   if (s.clock == s.u.decode.guardEnd)
      m.thr = guardDev;

   if (s.clock > s.u.decode.waitingEnd)
      return SYM_TIMEOUT;

   if (deep > c.maxDepth[3])
      return SYM_TIMEOUT;

   if (s.clock < m.winStart)
      return SYM_NONE;

   if (s0 < -m.thr && s0 < m.peak)
   {
      m.peak = s0;
      m.peakTime = s.clock;
      m.winEnd = s.clock + rt.p8;
   }

   if (s0 > m.thr && s0 > m.peak)
   {
      m.peak = s0;
      m.peakTime = s.clock;
      m.winEnd = s.clock + rt.p8;
   }

   if (s.clock != m.winEnd)
      return SYM_NONE;

   /* durations compare unsigned against the limits (int vs unsigned int in the reference) */
   if (m.stage == 0)
   {
      if (!m.symStart)
      {
         m.symStart = m.peakTime - rt.p1;
         m.winStart = m.peakTime + rt.p0;
         m.winEnd = m.winStart + rt.p1;
         m.peak = 0;
         m.peakTime = 0;
         return SYM_NONE;
      }

      m.symEnd = m.peakTime;

      uint32_t length = m.symEnd - m.symStart - rt.p1;

      if (m.peakTime == 0 || length < nfc_tu(c, 768 - 32) || length > nfc_tu(c, 768 + 32))
      {
         m.stage = 0; m.winStart = 0; m.winEnd = 0; m.symStart = 0; m.symEnd = 0;
         return SYM_NONE;
      }

      m.stage = 1;
      m.winStart = m.peakTime + rt.p1 - rt.p2;
      m.winEnd = m.winStart + rt.p1;
      m.peak = 0;
      m.peakTime = 0;
      return SYM_NONE;
   }

   if (m.stage == 1)
   {
      uint32_t length = m.peakTime - m.symEnd;

      if (m.peakTime == 0 || length < nfc_tu(c, 256 - 32) || length > nfc_tu(c, 256 + 32))
      {
         m.stage = 0; m.winStart = 0; m.winEnd = 0; m.symStart = 0; m.symEnd = 0;
         return SYM_NONE;
      }

      m.symEnd = m.peakTime;
      m.sync = m.symEnd + rt.p0;
      m.winStart = m.sync - rt.p4;
      m.winEnd = m.sync + rt.p4;
      m.thr = m.peak * 0.25f;
      m.c0 = 0;
      m.c1 = 0;
      m.peakTime = 0;
      m.peak = 0;

      s.u.decode.symValue = 0;
      s.u.decode.symStart = m.symStart - rt.delay;
      s.u.decode.symEnd = m.symEnd - rt.delay;
      s.u.decode.symPattern = V_S;
      return V_S;
   }

   return SYM_NONE;
}

NFC_DEV uint32_t nfcv_listen_symbol(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   float s0 = nfcv_burst_correlation(s, mem, m, taps);
   float sd = nfc_abs(s0);

   if (s.clock < m.winStart)
      return SYM_NONE;

   if (sd > m.thr && sd > m.peak)
   {
      m.c0 = s0;
      m.c1 = -s0;
      m.peak = sd;
      m.symEnd = s.clock;
   }

   if (s.clock != m.winEnd)
      return SYM_NONE;

   if (m.peak < m.thr)
      return V_S;

   m.symStart = m.symEnd;
   m.symEnd = m.symStart + rt.p0;
   m.sync = m.symEnd;
   m.winStart = m.sync - rt.p4;
   m.winEnd = m.sync + rt.p4;
   m.thr = m.peak * 0.25f;
   m.peakTime = 0;
   m.peak = 0;

   s.u.decode.symValue = m.c0 > m.c1 ? 0u : 1u;
   s.u.decode.symStart = m.symStart - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.delay;
   s.u.decode.symPattern = s.u.decode.symValue ? V_1 : V_0;

   return s.u.decode.symPattern;
}

NFC_DEV void nfcv_decode(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now, const NfcDecTaps &taps)
{

   if (s.u.decode.frameType == NFC_FRAME_POLL)
   {
      uint32_t pattern = nfcv_poll_symbol(c, s, mem, taps);

      if (pattern <= SYM_TIMEOUT)
         return;

      bool frameEnd = false, truncated = false, streamError = false;

      if (pattern == V_S)
         frameEnd = true;
      else if (pattern == V_E)
         streamError = true;
      else if (s.u.decode.bsBytes == s.u.decode.maxFrame)
         truncated = true;

      if (frameEnd || streamError || truncated)
      {
         if (s.u.decode.bsBytes > 0)
         {
            if (s.u.decode.bsBits == 8)
               nfc_push_byte(mem, s, s.u.decode.bsData);

            s.u.decode.frameEnd = s.u.decode.symEnd;

            nfc_pend_frame(s, NFC_FRAME_POLL, (truncated || streamError) ? NFC_FLAG_TRUNCATED : 0);
            return;
         }

         nfcv_reset(c, s, mem);
         return;
      }

      if (s.u.decode.bsBits == 8)
      {
         nfc_push_byte(mem, s, s.u.decode.bsData);
         s.u.decode.bsData = 0;
         s.u.decode.bsBits = 0;
      }

      s.u.decode.bsData |= (s.u.decode.symValue << s.u.decode.bsBits);
      s.u.decode.bsBits += s.u.decode.pulseCode ? 8u : 2u;
      return;
   }

   if (s.u.decode.frameType != NFC_FRAME_LISTEN)
      return;

   if (!s.u.decode.frameStart)
   {
      uint32_t pattern = nfcv_listen_start(c, s, mem, now, taps);

      if (pattern == V_S || pattern == SYM_TIMEOUT)
         nfc_wait_ended(mem, s, 3u, pattern == SYM_TIMEOUT && s.clock > s.u.decode.waitingEnd);

      if (pattern == V_S)
         s.u.decode.frameStart = s.u.decode.symStart;
      else if (pattern == SYM_TIMEOUT)
         nfcv_reset(c, s, mem);

      return;
   }

   uint32_t pattern = nfcv_listen_symbol(c, s, mem, taps);

   if (pattern <= SYM_TIMEOUT)
      return;

   bool frameEnd = false, truncated = false, streamError = false;

   if (pattern == V_S)
      frameEnd = true;
   else if (pattern == V_E)
      streamError = true;
   else if (s.u.decode.bsBytes == s.u.decode.maxFrame)
      truncated = true;

   if (frameEnd || streamError || truncated)
   {
      if (s.u.decode.bsBytes > 0)
      {
         if (s.u.decode.bsBits == 8)
            nfc_push_byte(mem, s, s.u.decode.bsData);

         s.u.decode.frameEnd = s.u.decode.symEnd;

         nfc_pend_frame(s, NFC_FRAME_LISTEN, (truncated || streamError) ? NFC_FLAG_TRUNCATED : 0);
         return;
      }

      nfcv_reset(c, s, mem);
      return;
   }

   if (s.u.decode.bsBits == 8)
   {
      nfc_push_byte(mem, s, s.u.decode.bsData);
      s.u.decode.bsData = 0;
      s.u.decode.bsBits = 0;
   }

   s.u.decode.bsData |= (s.u.decode.symValue << s.u.decode.bsBits);
   s.u.decode.bsBits++;
}

#endif
