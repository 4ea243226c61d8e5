/*
 * nfc_tech_f.hpp — JIS X 6319-4 / NFC-F (FeliCa): Manchester ASK at 212/424 kbps, both directions.
 *
 * Reference behaviour being matched: src/nfc-lib/lib-lab/lab-radio/src/main/cpp/tech/NfcF.cpp
 *   detectModulation 206-408 (>= 94 preamble pulses, 48-symbol preamble length, polarity),
 *   decodePollFrame 428-529, decodeListenFrame 534-636, decodePollFrameSymbolAsk 641-744,
 *   decodeListenFrameStartAsk 749-936, decodeListenFrameSymbolAsk 941-1042,
 *   resetModulation 1047-1071, process/processREQC/processOther 1076-1210, checkCrc 1215-1226.
 * Included by nfc_core.hpp (device code).
 */
#ifndef NFC_AMD_TECH_F_HPP
#define NFC_AMD_TECH_F_HPP

NFC_DEV void nfcf_protocol_defaults(const NfcConfig &c, NfcTiming &t)
{
   t.maxFrameSize = 256;
   t.protoGuardTime = nfc_tu(c, 1024);            /* NFCF_FGT_DEF */
   t.protoWaitingTime = nfc_tu(c, 256 * 16 * 16); /* NFCF_FWT_DEF */
}

NFC_DEV void nfcf_reset(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem)
{
   nfc_leave_lock(s, NFC_TECH_F);
}

/* CRC-16/XMODEM, big-endian on the wire */
NFC_DEV bool nfcf_crc_ok(const uint8_t *data, uint32_t len)
{
   if (len < 2)
      return false;

   uint32_t crc = nfc_crc16(data, len - 2, 0x0000u, false);
   uint32_t res = ((uint32_t)data[len - 2] << 8) | (uint32_t)data[len - 1];
   return res == crc;
}

NFC_DEV void nfcf_process(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, uint32_t type, const uint8_t *data, uint32_t len,
                          uint32_t &flags, uint32_t &phase)
{
   NfcTiming &t = mem.cold->tim[2];
   const bool poll = (type == NFC_FRAME_POLL);
   const bool crcOk = nfcf_crc_ok(data, len);

   t.guardTime = t.protoGuardTime;
   if (poll)
   {
      t.waitingTime = t.protoWaitingTime;
      nfc_wait_from_proto(mem, 2u);
   }

   /* REQC (command code 0x00 in the second byte) and its responses */
   if (poll && nfc_byte(data, len, 1) == 0x00)
   {
      t.lastCommand = 0x00, nfc_command_written(mem, 2u);

      int tsn = (int)nfc_byte(data, len, 5);

      t.maxFrameSize = 256;
      t.protoGuardTime = nfc_tu(c, 1024);
      t.protoWaitingTime = nfc_tu(c, 256 * 16 * 16);
      nfc_wait_proto_written(mem, 2u);
      t.guardTime = nfc_tu(c, 1024);
      t.waitingTime = (uint32_t)(c.stu * (double)(512 * 64 + (tsn + 1) * 256 * 64)); /* FDT_ATQC + slots */
      nfc_wait_overridden(mem, 2u);

      phase = NFC_PHASE_SELECTION;
      if (!crcOk)
         flags |= NFC_FLAG_CRC;
   }
   else if (!poll && t.lastCommand == 0x00)
   {
      phase = NFC_PHASE_SELECTION;
      if (!crcOk)
         flags |= NFC_FLAG_CRC;
   }
   else
   {
      phase = NFC_PHASE_APPLICATION;
      if (!crcOk)
         flags |= NFC_FLAG_CRC;
   }

   const bool locked = (s.lockTech == NFC_TECH_F);
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
      t.lastCommand = 0, nfc_command_written(mem, 2u);
   }

   s.u.decode.frameStart = 0;
   s.u.decode.frameEnd = 0;
}

/* the preamble tracker shared by search (NfcF.cpp:267-405) and listen-SOF (NfcF.cpp:815-933);
 * returns true when a complete, length-checked preamble has just ended */
template <class M>
NFC_DEV bool nfcf_track_preamble(NfcStreamState &s, M &m, const NfcRate &rt, float sd, float s0, bool above, uint32_t &polarity, uint32_t *cleared,
                                 uint32_t *used = nullptr, uint32_t usedBit = 0u, NfcFBound *bound = nullptr)
{
   if (above)
   {
      if (sd > m.peak)
      {
         m.peak = sd;
         m.peakTime = s.clock;

         if (!m.sync)
         {
            m.syncValue = sd;
            m.c0 = s0;
            m.winEnd = s.clock + rt.p8;
         }
      }
   }

   if (s.clock == m.sync)
   {
      m.syncValue = sd;
      m.lastValue = s0;
   }

   if (s.clock != m.winEnd)
      return false;

   /* the pulse counter and the threshold of the last pulse are looked at from here on: if the record has not started
    * over since the lane began, what the lane inherited matters (NfcStreamCold::usedTech) */
   if (used && cleared && !*cleared)
   {
      *used |= usedBit;

      /* ... and what this evaluation requires of them (NfcFBound): the counter on the same side of 94; the threshold,
       * while it is still the one the lane was given, on the same side of the pulse */
      if (bound)
      {
         const uint32_t counted = m.pulses;

         if (counted < 94u)
            bound->lowMax = counted + 1u > bound->lowMax ? counted + 1u : bound->lowMax;
         else
            bound->highMin = (bound->highMin == 0u || counted + 1u < bound->highMin) ? counted + 1u : bound->highMin;

         if (!(bound->flags & NFC_FBOUND_THR_OWN))
         {
            if (counted < 94u && m.peakTime == 0)
               ; /* cleared whatever the threshold */
            else if (counted < 94u && m.syncValue < m.thr)
            {
               bound->thrAbove = (bound->flags & NFC_FBOUND_ABOVE) && bound->thrAbove > m.syncValue ? bound->thrAbove : m.syncValue;
               bound->flags |= NFC_FBOUND_ABOVE;
            }
            else if (m.syncValue > m.thr)
            {
               bound->thrBelow = (bound->flags & NFC_FBOUND_BELOW) && bound->thrBelow < m.syncValue ? bound->thrBelow : m.syncValue;
               bound->flags |= NFC_FBOUND_BELOW;
            }
            else
               bound->flags |= NFC_FBOUND_EXACT; /* (equal to the threshold, or not a number: the end-of-preamble path) */
         }
         else if (!(m.syncValue > m.thr) && !(counted < 94u && (m.peakTime == 0 || m.syncValue < m.thr)))
            bound->flags |= NFC_FBOUND_EXACT; /* the end-of-preamble path reads more of the record than the bounds speak of */
      }
   }

   if (m.pulses++ < 94)
   {
      if (m.peakTime == 0 || m.syncValue < m.thr)
      {
         m.symStart = 0; m.symEnd = 0; m.sync = 0; m.syncValue = 0; m.winStart = 0; m.winEnd = 0;
         m.pulses = 0; m.thr = 0; m.peak = 0; m.peakTime = 0;
         if (cleared)
            *cleared = 1; /* the pulse counter starts over (NfcStreamCold::clearedF) */
         if (bound)
            bound->flags |= NFC_FBOUND_THR_OWN;
         return false;
      }
   }

   if (m.syncValue > m.thr)
   {
      if (bound)
         bound->flags |= NFC_FBOUND_THR_OWN; /* (set from this pulse below) */

      if (!m.symStart)
         m.symStart = m.peakTime - rt.p2;

      m.symEnd = m.peakTime;
      m.sync = m.symEnd + rt.p2;
      m.winStart = m.sync - rt.p8;
      m.winEnd = m.sync + rt.p8;
      m.thr = m.peak / 2;
      m.lastPhase = m.lastValue;
      m.peakTime = 0;
      m.peak = 0;
      return false;
   }

   if ((m.lastPhase < 0 && m.c0 < 0) || (m.lastPhase > 0 && m.c0 > 0))
      m.symStart -= rt.p2;

   int length = (int)(m.symEnd - m.symStart);
   int minimum = (int)(rt.preamble - rt.p4);
   int maximum = (int)(rt.preamble + rt.p4);

   if (length < minimum || length > maximum)
   {
      m.symStart = 0; m.symEnd = 0; m.sync = 0; m.syncValue = 0; m.winStart = 0; m.winEnd = 0;
      m.pulses = 0; m.thr = 0; m.peak = 0; m.peakTime = 0;
      if (cleared)
         *cleared = 1;
      if (bound)
         bound->flags |= NFC_FBOUND_THR_OWN;
      return false;
   }

   polarity = m.lastPhase > 0 ? 0u : 1u; /* observed / reversed polarity (the reference's searchModeState) */
   m.sync = m.sync + rt.p2;
   m.winStart = m.sync - rt.p4;
   m.winEnd = m.sync + rt.p4;
   m.peakTime = 0;
   m.peak = 0;

   return true;
}

/* history reads of the two preamble correlators */
struct NfcTapsF
{
   NfcTap t[3];
};

NFC_DEV void nfcf_load_taps(const NfcConfig &c, const NfcStreamState &s, const NfcLaneMem &mem, NfcTapsF &taps)
{
   taps.t[1] = nfc_tap_raw(mem, s.clock, c.f[1], c.corrOffset[3], s.posF[0], false);
   taps.t[2] = nfc_tap_raw(mem, s.clock, c.f[2], c.corrOffset[4], s.posF[1], false);
}

template <int R>
NFC_DEV bool nfcf_detect_decide(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float s0, float num, float deep, float minimumCorrelation);

template <int R>
NFC_DEV bool nfcf_detect_rate(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcTapsF &taps, const NfcNow &now,
                              float minimumCorrelation)
{
   const NfcRate &rt = c.f[R];
   NfcDetF &m = s.u.search.detF[R - 1];

   /* NFC-F correlates the undelayed signal: the entering sample and its depth are the current ones */
   NfcTap tap = taps.t[R];
   tap.in = now.x;
   tap.c3 = nfc_previous_sum(mem, s, m, rt, c.corrOffset[2 + R], s.posF[R - 1]);

   NfcCorr k = nfc_corr_apply(mem, m, tap, c.corrOffset[2 + R], s.posF[R - 1]);

   return nfcf_detect_decide<R>(c, s, mem, k.s0, k.s0 - k.s1, now.depth, minimumCorrelation);
}

/* what the detector does with this sample's correlation (s0, num = S0 - S1) and modulation depth; the correlator itself
 * has been stepped by the caller */
template <int R>
NFC_DEV bool nfcf_detect_decide(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float s0, float num, float deep, float minimumCorrelation)
{
   const NfcRate &rt = c.f[R];
   NfcDetF &m = s.u.search.detF[R - 1];

   /* one branch for the common case: without a reset, with the window closed or with a correlation that cannot exceed
    * the threshold away from the synchronisation point and the window end, nothing below changes the record */
   const bool reset = deep > c.maxDepth[2] || (m.peakTime && s.clock > m.peakTime + rt.p1);
   const bool eventful = s.clock >= m.winStart &&
                         (nfc_may_exceed(num, (float)rt.p2, minimumCorrelation) || s.clock == m.sync || s.clock == m.winEnd);

   /* from a reset on the record is the lane's own (NfcStreamCold::usedTech) */
   if (reset && mem.linked)
      *mem.flags |= 1u << (15 + R);

   /* (a reset of a record that is clear already - every sample of a 100 % ASK pause asks for one - changes nothing) */
   const bool clear = (m.symStart | m.symEnd | m.winStart | m.winEnd | m.sync | m.peakTime | nfc_bits(m.peak)) == 0u;

   if ((!reset || clear) && !eventful)
      return false;

   if (reset)
   {
      /* (detectorPeak* are never set by this detector: nothing to clear) */
      m.symStart = 0; m.symEnd = 0; m.winStart = 0; m.winEnd = 0; m.sync = 0;
      m.peakTime = 0; m.peak = 0;
   }

   if (s.clock < m.winStart)
      return false;

   /* the quotient is needed when it can exceed the threshold, and at the synchronisation sample (captured) */
   float sd = 0.0f;
   if (nfc_may_exceed(num, (float)rt.p2, minimumCorrelation) || s.clock == m.sync)
      sd = nfc_abs(num) / (float)rt.p2;

   uint32_t polarity = 0;

   /* the tracker looks at the record: if the lane has not cleared it since it started, what it inherited matters */
   if (mem.linked && !((*mem.flags >> (15 + R)) & 1u))
      *mem.flags |= 1u << (13 + R);

   if (!nfcf_track_preamble(s, m, rt, sd, s0, sd > minimumCorrelation, polarity, &mem.cold->clearedF[R - 1], mem.linked ? mem.flags : nullptr, 1u << (11 + R),
                            mem.linked ? &mem.cold->boundF[R - 1] : nullptr))
      return false;

   /* (a preamble completed on a pulse memory the lane was given: the lock copies the counter - only the assumed values will do) */
   if (mem.linked && !mem.cold->clearedF[R - 1])
      mem.cold->boundF[R - 1].flags |= NFC_FBOUND_EXACT;

   /* preamble complete: lock this bitrate, the sync bytes follow (copy the detector record before it is parked) */
   const NfcDetF k0 = m;
   const uint32_t pos = s.posF[R - 1];

   NfcDecodeRegs &out = nfc_take_lock(mem, rt, (uint32_t)R, c.corrOffset[2 + R], pos);
   NfcMod &d = out.lock;
   d.stage = polarity;
   d.winStart = k0.winStart; d.winEnd = k0.winEnd; d.sync = k0.sync; d.pulses = k0.pulses;
   d.thr = k0.thr; d.lastPhase = k0.lastPhase; d.lastValue = k0.lastValue; d.syncValue = k0.syncValue; d.c0 = k0.c0;
   d.symStart = k0.symStart; d.symEnd = k0.symEnd; d.acc = k0.acc; d.peak = k0.peak; d.peakTime = k0.peakTime;

   out.symStart = k0.symStart;
   out.symEnd = k0.symEnd;
   out.symPattern = F_S;

   out.frameType = NFC_FRAME_POLL;
   out.frameRate = rt.symbolsPerSecond;
   out.frameStart = k0.symStart;
   out.frameEnd = 0;

   return true;
}

/* the caller has checked that the search bank is armed (nfc_search_detect) */
NFC_DEV bool nfcf_detect(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcTapsF &taps, const NfcNow &now)
{
   const float minimumCorrelation = s.env * c.corrThreshold[2];

   if (nfcf_detect_rate<1>(c, s, mem, taps, now, minimumCorrelation))
      return true;

   return nfcf_detect_rate<2>(c, s, mem, taps, now, minimumCorrelation);
}

/* Manchester data symbols, identical for both directions (NfcF.cpp:641-744 and 941-1042) */
NFC_DEV uint32_t nfcf_data_symbol(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   const uint32_t lockPos = nfc_lock_pos(s);
   NfcTap tap;
   tap.in = taps.x0; tap.out = taps.x2; tap.c2 = taps.c2; tap.c3 = taps.c3;
   NfcCorr k = nfc_corr_apply(mem, m, tap, s.u.decode.lockBase, lockPos);
   float sd = nfc_abs(k.s0 - k.s1) / (float)rt.p2;

   if (s.clock < m.winStart)
      return SYM_NONE;

   if (sd > m.thr && sd > m.peak)
   {
      m.peak = sd;
      m.peakTime = s.clock;
   }

   if (s.clock == m.sync)
   {
      m.c0 = k.s0;
      m.c1 = k.s1;
   }

   if (s.clock != m.winEnd)
      return SYM_NONE;

   if (!m.peakTime)
      return F_E;

   m.symStart = m.symEnd;
   m.symEnd = m.peakTime;
   m.sync = m.symEnd + rt.p1;
   m.winStart = m.sync - rt.p4;
   m.winEnd = m.sync + rt.p4;
   m.thr = m.peak / 2;
   m.peakTime = 0;
   m.peak = 0;

   s.u.decode.symStart = m.symStart - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.delay;

   if ((m.stage == 0 && m.c0 > m.c1) || (m.stage == 1 && m.c0 < m.c1))
   {
      s.u.decode.symValue = 0;
      s.u.decode.symPattern = F_L;
   }
   else
   {
      s.u.decode.symValue = 1;
      s.u.decode.symPattern = F_H;
   }

   return s.u.decode.symPattern;
}

NFC_DEV uint32_t nfcf_listen_start(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   const uint32_t base = s.u.decode.lockBase;
   const uint32_t pos = nfc_lock_pos(s);

   NfcTap tap;
   tap.in = taps.x0; tap.out = taps.x2; tap.c2 = taps.c2; tap.c3 = taps.c3;
   const float guardDev = taps.m0;

   /* the box sum runs from the end of the poll frame, the ring only from one symbol before the guard */
   m.acc += tap.in;
   m.acc -= tap.out;

   if (s.clock < (uint32_t)(s.u.decode.guardEnd - rt.p1))
      return SYM_NONE;

   NFC_AT(mem, NFC_R_CORR, base + pos) = m.acc;

   float s0 = m.acc - tap.c2;
   float s1 = tap.c2 - tap.c3;
   float sd = nfc_abs(s0 - s1) / (float)rt.p2;

   if (s.clock < s.u.decode.guardEnd)
      return SYM_NONE;

   if (s.clock == s.u.decode.guardEnd)
      m.thr = guardDev * 10.0f;

   if (s.clock > s.u.decode.waitingEnd)
      return SYM_TIMEOUT;

   if (s.clock < m.winStart)
      return SYM_NONE;

   uint32_t polarity = 0;

   if (!nfcf_track_preamble(s, m, rt, sd, s0, sd >= m.thr, polarity, nullptr))
      return SYM_NONE;

   m.stage = polarity;

   s.u.decode.symStart = m.symStart - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.delay;
   s.u.decode.symPattern = F_S;

   return F_S;
}

/* bits are MSB first, no parity; frame = 2 sync bytes + payload */
NFC_DEV void nfcf_frame(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, uint32_t pattern, uint32_t type)
{
   bool frameEnd = false, truncated = false;

   if (pattern == F_E)
      frameEnd = true;
   else if (s.u.decode.bsBytes == s.u.decode.maxFrame)
      truncated = true;

   if (frameEnd || truncated)
   {
      if (s.u.decode.bsBytes > 2)
      {
         s.u.decode.frameEnd = s.u.decode.symEnd;

         nfc_pend_frame(s, type, truncated ? NFC_FLAG_TRUNCATED : 0);
         return;
      }

      nfcf_reset(c, s, mem);
      return;
   }

   s.u.decode.bsData = (s.u.decode.bsData << 1) | s.u.decode.symValue;

   if (++s.u.decode.bsBits == 8)
   {
      nfc_push_byte(mem, s, s.u.decode.bsData);
      s.u.decode.bsData = 0;
      s.u.decode.bsBits = 0;
   }
}

NFC_DEV void nfcf_decode(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now, const NfcDecTaps &taps)
{
   if (s.u.decode.frameType == NFC_FRAME_POLL)
   {
      uint32_t pattern = nfcf_data_symbol(c, s, mem, taps);

      if (pattern > SYM_TIMEOUT)
         nfcf_frame(c, s, mem, pattern, NFC_FRAME_POLL);

      return;
   }

   if (s.u.decode.frameType != NFC_FRAME_LISTEN)
      return;

   if (!s.u.decode.frameStart)
   {
      uint32_t pattern = nfcf_listen_start(c, s, mem, taps);

      if (pattern == F_S || pattern == SYM_TIMEOUT)
         nfc_wait_ended(mem, s, 2u, pattern == SYM_TIMEOUT && s.clock > s.u.decode.waitingEnd);

      if (pattern == F_S)
         s.u.decode.frameStart = s.u.decode.symStart;
      else if (pattern == SYM_TIMEOUT)
         nfcf_reset(c, s, mem);

      return;
   }

   uint32_t pattern = nfcf_data_symbol(c, s, mem, taps);

   if (pattern > SYM_TIMEOUT)
      nfcf_frame(c, s, mem, pattern, NFC_FRAME_LISTEN);
}

#endif
