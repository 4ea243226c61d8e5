/*
 * nfc_tech_a.hpp — ISO14443-A / NFC-A: modified-Miller ASK poll frames at 106/212/424 kbps,
 * Manchester-OOK (106k) and BPSK (212k/424k) listen frames.
 *
 * Reference behaviour being matched: src/nfc-lib/lib-lab/lab-radio/src/main/cpp/tech/NfcA.cpp
 *   detectModulation 217-411, decodePollFrame 432-563, decodeListenFrame 568-803,
 *   symbol decoders 812-1421, resetFrameSearch/resetModulation 1426-1475, process* 1480-1973,
 *   checkCrc/checkParity 1978-2005.
 * Included by nfc_core.hpp (device code).
 */
#ifndef NFC_AMD_TECH_A_HPP
#define NFC_AMD_TECH_A_HPP

NFC_DEV void nfca_protocol_defaults(const NfcConfig &c, NfcTiming &t)
{
   t.maxFrameSize = 256;
   t.protoGuardTime = nfc_tu(c, 1024);             /* NFCA_FGT_DEF */
   t.protoWaitingTime = nfc_tu(c, 256 * 16 * 16);  /* NFCA_FWT_DEF */
}

/* resetModulation, NfcA.cpp:1451-1475 */
NFC_DEV void nfca_reset(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem)
{
   nfc_leave_lock(s, NFC_TECH_A);
}

/* resetFrameSearch, NfcA.cpp:1426-1446 */
NFC_DEV void nfca_reset_search(NfcStreamState &s, NfcMod &m)
{
   m.symStart = 0; m.symEnd = 0; m.symRise = 0;
   m.sync = 0; m.winStart = 0; m.winEnd = 0; m.pulses = 0;
   m.peakTime = 0; m.peak = 0; m.auxTime = 0; m.aux = 0;
   s.u.decode.frameStart = 0;
}

NFC_DEV bool nfca_crc_ok(const uint8_t *data, uint32_t len)
{
   if (len < 2)
      return true;

   uint32_t crc = nfc_crc16(data, len - 2, 0x6363u, true);
   uint32_t res = (uint32_t)data[len - 2] | ((uint32_t)data[len - 1] << 8);
   return res == crc;
}

/* odd parity check as NfcA.cpp:1994-2005: returns the parity bit xor-ed with every set data bit */
NFC_DEV uint32_t nfca_parity(uint32_t value, uint32_t parity)
{
   /* parity of the low byte, folded: same result as flipping the bit once per set data bit */
   uint32_t v = value & 0xFFu;
   v ^= v >> 4;
   v ^= v >> 2;
   v ^= v >> 1;
   return parity ^ (v & 1u);
}

NFC_DEV void nfca_default_timing(const NfcConfig &c, NfcTiming &t)
{
   t.maxFrameSize = 256;
   t.protoGuardTime = nfc_tu(c, 1024);
   t.protoWaitingTime = nfc_tu(c, 256 * 16 * 16);
}

/* frame classification + protocol timing feedback, NfcA.cpp:1480-1973 */
NFC_DEV void nfca_process(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, uint32_t type,
                          const uint8_t *data, uint32_t len, uint32_t &flags, uint32_t &phase)
{
   NfcTiming &t = mem.cold->tim[0];
   const bool poll = (type == NFC_FRAME_POLL);
   const uint32_t b0 = nfc_byte(data, len, 0);
   const bool crcOk = nfca_crc_ok(data, len); /* one CRC pass per frame, used by whichever classification applies */

   if (poll)
   {
      t.guardTime = t.protoGuardTime;
      t.waitingTime = t.protoWaitingTime;
      nfc_wait_from_proto(mem, 0u);
   }
   else
   {
      t.guardTime = t.protoGuardTime;
   }

   bool done = false;

   /* REQA / WUPA */
   if (poll)
   {
      if ((b0 == 0x26 || b0 == 0x52) && len == 1)
      {
         phase = NFC_PHASE_SELECTION;
         t.lastCommand = b0, nfc_command_written(mem, 0u);
         nfca_default_timing(c, t);
         nfc_wait_proto_written(mem, 0u);
         t.guardTime = nfc_tu(c, 1024);     /* NFCA_FGT_DEF  */
         t.waitingTime = nfc_tu(c, 128 * 18); /* NFCA_FWT_ATQA */
         nfc_wait_overridden(mem, 0u);
         s.chainedA = 0;
         done = true;
      }
   }
   else if (t.lastCommand == 0x26 || t.lastCommand == 0x52)
   {
      phase = NFC_PHASE_SELECTION;
      done = true;
   }

   /* HLTA */
   if (!done && poll && b0 == 0x50 && len == 4 && !(flags & NFC_FLAG_CRC))
   {
      phase = NFC_PHASE_SELECTION;
      if (!crcOk)
         flags |= NFC_FLAG_CRC;
      t.lastCommand = b0, nfc_command_written(mem, 0u);
      nfca_default_timing(c, t);
      nfc_wait_proto_written(mem, 0u);
      nfc_wait_overridden(mem, 0u); /* (no wait follows: the decoder resets) */
      s.chainedA = 0;
      nfca_reset(c, s, mem);
      done = true;
   }

   if (!done)
   {
      if (!(s.chainedA & NFC_FLAG_ENCRYPTED))
      {
         const uint32_t last = t.lastCommand;

         /* SEL1/2/3 */
         if (poll ? (b0 == 0x93 || b0 == 0x95 || b0 == 0x97) : (last == 0x93 || last == 0x95 || last == 0x97))
         {
            phase = NFC_PHASE_SELECTION;
            if (poll)
            {
               t.lastCommand = b0, nfc_command_written(mem, 0u);
               t.guardTime = nfc_tu(c, 1024);
               t.waitingTime = nfc_tu(c, 128 * 18);
               nfc_wait_overridden(mem, 0u);
            }
         }
         /* RATS */
         else if (poll ? (b0 == 0xE0) : (last == 0xE0))
         {
            if (poll)
            {
               static const uint16_t fsd[16] = {16, 24, 32, 40, 48, 64, 96, 128, 256, 512, 1024, 2048, 4096, 0, 0, 0};
               uint32_t fsdi = (nfc_byte(data, len, 1) >> 4) & 0x0F;
               t.lastCommand = b0, nfc_command_written(mem, 0u);
               t.maxFrameSize = fsd[fsdi];
               t.waitingTime = nfc_tu(c, 71680); /* NFC_FWT_ACTIVATION */
               nfc_wait_overridden(mem, 0u);
            }
            else
            {
               uint32_t offset = 0;
               uint32_t tl = nfc_byte(data, len, offset++);

               if (tl > 0)
               {
                  uint32_t t0 = nfc_byte(data, len, offset++);

                  if (t0 & 0x10)
                     offset++;

                  if (t0 & 0x20)
                  {
                     uint32_t tb = nfc_byte(data, len, offset++);
                     uint32_t fwi = (tb >> 4) & 0x0f;

                     if (fwi == 15)
                        fwi = 4;

                     t.protoWaitingTime = nfc_tu(c, 4096 << fwi); /* NFC_FWT_TABLE[fwi] */
                  }
                  else
                  {
                     t.protoWaitingTime = nfc_tu(c, 256 * 16 * 16);
                  }

                  nfc_wait_proto_written(mem, 0u);
               }
            }

            phase = NFC_PHASE_SELECTION;
            if (!crcOk)
               flags |= NFC_FLAG_CRC;
         }
         /* PPS */
         else if (poll ? ((b0 & 0xF0) == 0xD0) : (last == 0xD0))
         {
            if (poll)
               t.lastCommand = b0 & 0xF0, nfc_command_written(mem, 0u);
            phase = NFC_PHASE_SELECTION;
            if (!crcOk)
               flags |= NFC_FLAG_CRC;
         }
         /* Mifare AUTH */
         else if (poll ? (b0 == 0x60 || b0 == 0x61) : (last == 0x60 || last == 0x61))
         {
            phase = NFC_PHASE_APPLICATION;
            if (poll)
            {
               t.lastCommand = b0, nfc_command_written(mem, 0u);
               if (!crcOk)
                  flags |= NFC_FLAG_CRC;
            }
            else
            {
               s.chainedA = NFC_FLAG_ENCRYPTED;
            }
         }
         /* I-Block */
         else if (poll ? ((b0 & 0xE2) == 0x02 && len > 4) : (last == 0x02))
         {
            if (poll)
               t.lastCommand = b0 & 0xE2, nfc_command_written(mem, 0u);
            phase = NFC_PHASE_APPLICATION;
            if (!crcOk)
               flags |= NFC_FLAG_CRC;
         }
         /* R-Block */
         else if (poll ? ((b0 & 0xE6) == 0xA2 && len == 3) : (last == 0xA2))
         {
            if (poll)
               t.lastCommand = b0 & 0xE6, nfc_command_written(mem, 0u);
            phase = NFC_PHASE_APPLICATION;
            if (!crcOk)
               flags |= NFC_FLAG_CRC;
         }
         /* S-Block */
         else if (poll ? ((b0 & 0xC7) == 0xC0 && len == 4) : (last == 0xC0))
         {
            if (poll)
               t.lastCommand = b0 & 0xC7, nfc_command_written(mem, 0u);
            phase = NFC_PHASE_APPLICATION;
            if (!crcOk)
               flags |= NFC_FLAG_CRC;
         }
         else
         {
            phase = NFC_PHASE_APPLICATION;
            if (!crcOk)
               flags |= NFC_FLAG_CRC;
         }
      }
      else
      {
         /* everything after AUTH is ciphered: parity is meaningless, frame is application level */
         flags &= ~(uint32_t)NFC_FLAG_PARITY;
         phase = NFC_PHASE_APPLICATION;
      }
   }

   flags |= s.chainedA;

   const bool locked = (s.lockTech == NFC_TECH_A);
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
      t.lastCommand = 0, nfc_command_written(mem, 0u);
   }

   if (locked)
   {
      s.u.decode.frameStart = 0;
      s.u.decode.frameEnd = 0;
   }
}

/* history reads of the three poll-SOF correlators, issued before the front end stores the new sample */
struct NfcTapsA
{
   NfcTap t[3];
   float deep[3]; /* modulation depth one eighth of a symbol before the (delayed) sample: used while a pause is tracked */
};

template <int R>
NFC_DEV void nfca_load_taps_rate(const NfcConfig &c, const NfcStreamState &s, const NfcLaneMem &mem, NfcTapsA &taps)
{
   /* ring[(idx - 1) % p1] is read in nfca_detect_rate only after a gap in the search (see bankClock) */
   taps.t[R] = nfc_tap_raw(mem, s.clock, c.a[R], c.corrOffset[R], s.posA[R], false);

   /* read with the others although only needed during a pause: a load inside the detector is waited for with
    * everything else outstanding, the ring stores of the detectors before it included (p8 >= 1: never the slot being
    * written) */
   taps.deep[R] = NFC_AT(mem, NFC_R_DEPTH, (s.clock - c.a[R].delay - c.a[R].p8) & NFC_FMASK);
}

NFC_DEV void nfca_load_taps(const NfcConfig &c, const NfcStreamState &s, const NfcLaneMem &mem, NfcTapsA &taps)
{
   nfca_load_taps_rate<0>(c, s, mem, taps);
   nfca_load_taps_rate<1>(c, s, mem, taps);
   nfca_load_taps_rate<2>(c, s, mem, taps);
}

/* ---- search: SOF of a poll frame = one modified-Miller pause, NfcA.cpp:217-411 ---- */
/* what the detector does with the correlation of this sample (num = S0 - S1 of its box-sum correlator, deep = modulation
 * depth one eighth of a symbol before the delayed sample); the correlator itself has been stepped by the caller */
template <int R>
NFC_DEV bool nfca_detect_decide(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float num, float deepTap, float minimumCorrelation,
                                float minimumDepth);

template <int R>
NFC_DEV bool nfca_detect_rate(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcTapsA &taps, const NfcNow &now,
                              float minimumCorrelation, float minimumDepth)
{
   const NfcRate &rt = c.a[R];
   NfcDetA &m = s.u.search.detA[R];

   NfcTap tap = taps.t[R];
   if (rt.delay == 0)
      tap.in = now.x; /* the newest sample is not in memory yet when the taps are read */

   tap.c3 = nfc_previous_sum(mem, s, m, rt, c.corrOffset[R], s.posA[R]);

   NfcCorr k = nfc_corr_apply(mem, m, tap, c.corrOffset[R], s.posA[R]);

   return nfca_detect_decide<R>(c, s, mem, k.s0 - k.s1, taps.deep[R], minimumCorrelation, minimumDepth);
}

template <int R>
NFC_DEV bool nfca_detect_decide(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, float num, float deepTap, float minimumCorrelation,
                                float minimumDepth)
{
   const NfcRate &rt = c.a[R];
   NfcDetA &m = s.u.search.detA[R];

   /* nothing below changes the record or returns true unless the tracked pause timed out, or the search window is
    * open and either the correlation can exceed the threshold or the window ends now: one branch for the common
    * case instead of one per condition */
   const bool timeout = m.peakTime && s.clock > m.peakTime + rt.p1;
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
      const float sd = num / (float)rt.p2;

      if (!m.symStart)
      {
         if (sd < -minimumCorrelation)
         {
            const float deep = deepTap;

            if (sd < m.peak)
            {
               m.peak = sd;
               m.peakTime = s.clock;
               m.winEnd = s.clock + rt.p4;
            }

            if (deep > m.aux)
               m.aux = deep;
         }
      }
      else if (sd > minimumCorrelation)
      {
         if (sd > m.peak)
         {
            m.peak = sd;
            m.peakTime = s.clock;
         }
      }
   }

   if (s.clock != m.winEnd)
      return false;

   if (!m.symStart)
   {
      if (m.aux < minimumDepth)
      {
         m.symStart = 0; m.winStart = 0; m.winEnd = 0;
         m.peakTime = 0; m.peak = 0; m.aux = 0;
         return false;
      }

      const uint32_t sync = m.peakTime + rt.p2;
      m.winStart = sync - rt.p8;
      m.winEnd = sync + rt.p8;
      m.symStart = m.peakTime - rt.p2;
      m.peakTime = 0;
      m.peak = 0;
      return false;
   }

   const uint32_t symEnd = m.peakTime;
   const uint32_t width = symEnd - m.symStart;
   const uint32_t minimumWidth = rt.p1 - rt.p4;
   const uint32_t maximumWidth = rt.p1 + rt.p4;

   if (m.peakTime == 0 || m.aux < minimumDepth || width < minimumWidth || width > maximumWidth)
   {
      m.symStart = 0; m.winStart = 0; m.winEnd = 0;
      m.peakTime = 0; m.peak = 0; m.aux = 0;
      return false;
   }

   /* SOF pause recognised: lock this bitrate (the detector record is about to be parked: take what is needed) */
   const uint32_t symStart = m.symStart, pos = s.posA[R];
   const float peak = m.peak, acc = m.acc, aux = m.aux;

   NfcDecodeRegs &out = nfc_take_lock(mem, rt, (uint32_t)R, c.corrOffset[R], pos);
   NfcMod &d = out.lock;
   d.symStart = symStart;
   d.symEnd = symEnd;
   d.pulses = width;
   d.sync = symEnd + rt.p1;
   d.winStart = d.sync - rt.p8;
   d.winEnd = d.sync + rt.p8;
   d.thr = peak / 2;
   d.acc = acc;
   d.aux = aux;

   out.frameType = NFC_FRAME_POLL;
   out.frameRate = rt.symbolsPerSecond;
   out.frameStart = symStart - rt.delay;
   out.frameEnd = 0;

   out.symValue = 0;
   out.symStart = symStart - rt.delay;
   out.symEnd = symEnd - rt.delay;
   out.symPattern = A_Z;

   return true;
}

/* the caller has checked that the search bank is armed (nfc_search_detect) */
NFC_DEV bool nfca_detect(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcTapsA &taps, const NfcNow &now)
{
   const float minimumCorrelation = s.env * c.corrThreshold[0];
   const float minimumDepth = c.minDepth[0];

   if (nfca_detect_rate<0>(c, s, mem, taps, now, minimumCorrelation, minimumDepth))
      return true;
   if (nfca_detect_rate<1>(c, s, mem, taps, now, minimumCorrelation, minimumDepth))
      return true;
   if (nfca_detect_rate<2>(c, s, mem, taps, now, minimumCorrelation, minimumDepth))
      return true;

   return false;
}

/* ---- poll frame symbols (modified Miller), NfcA.cpp:812-934 ---- */
NFC_DEV uint32_t nfca_poll_symbol(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   const uint32_t pos = nfc_lock_pos(s);
   NfcTap tap;
   tap.in = taps.x0; tap.out = taps.x2; tap.c2 = taps.c2; tap.c3 = taps.c3;
   NfcCorr k = nfc_corr_apply(mem, m, tap, s.u.decode.lockBase, pos);
   float sd = nfc_abs(k.s0 - k.s1) / (float)rt.p2;

   if (s.clock < m.winStart)
      return SYM_NONE;

   if (sd > m.peak && sd > m.thr)
   {
      m.peak = sd;
      m.peakTime = s.clock;
   }

   if (s.clock == m.sync)
   {
      m.cD = sd;
      m.c0 = k.s0;
      m.c1 = k.s1;
   }

   if (s.clock != m.winEnd)
      return SYM_NONE;

   if (m.cD < m.thr)
   {
      m.symStart = m.symEnd;
      m.symEnd = m.sync;
      m.symRise = m.symStart;
      s.u.decode.symValue = 1;
      s.u.decode.symPattern = A_Y;
   }
   else if (m.c0 > m.c1)
   {
      m.symStart = m.symEnd;
      m.symEnd = m.peakTime;
      m.symRise = m.peakTime - rt.p2;
      s.u.decode.symValue = 0;
      s.u.decode.symPattern = A_Z;
   }
   else
   {
      m.symStart = m.symEnd;
      m.symEnd = m.peakTime;
      m.symRise = m.peakTime;
      s.u.decode.symValue = 1;
      s.u.decode.symPattern = A_X;
   }

   m.sync = m.symEnd + rt.p1;
   m.winStart = m.sync - rt.p8;
   m.winEnd = m.sync + rt.p8;
   m.cD = 0; m.c0 = 0; m.c1 = 0;
   m.peakTime = 0;
   m.peak = 0;

   s.u.decode.symStart = m.symStart - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.delay;
   s.u.decode.symEdge = m.symRise - rt.delay;

   return s.u.decode.symPattern;
}

/* ---- poll frame assembly, NfcA.cpp:432-563 ---- */
NFC_DEV void nfca_poll_frame(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, uint32_t pattern)
{
   bool frameEnd = false, truncated = false;

   if (pattern == A_Y && (s.u.decode.bsPrevious == A_Y || s.u.decode.bsPrevious == A_Z))
      frameEnd = true;
   else if (s.u.decode.bsBytes == s.u.decode.maxFrame)
      truncated = true;

   if (frameEnd || truncated)
   {
      if (s.u.decode.bsBytes > 0 || s.u.decode.bsBits == 7)
      {
         if (s.u.decode.bsBits >= 7)
            nfc_push_byte(mem, s, s.u.decode.bsData);

         uint32_t flags = 0;

         if (s.u.decode.bsFlags & NFC_FLAG_PARITY)
            flags |= NFC_FLAG_PARITY;
         if (truncated)
            flags |= NFC_FLAG_TRUNCATED;
         if (s.u.decode.bsBytes == 1 && s.u.decode.bsBits == 7)
            flags |= NFC_FLAG_SHORT;

         nfc_pend_frame(s, NFC_FRAME_POLL, flags);
         return;
      }

      nfca_reset(c, s, mem);
      return;
   }

   if (s.u.decode.symEdge)
      s.u.decode.frameEnd = s.u.decode.symEdge;

   if (s.u.decode.bsPrevious)
   {
      uint32_t value = (s.u.decode.bsPrevious == A_X) ? 1u : 0u;

      if (s.u.decode.bsBits < 8)
      {
         s.u.decode.bsData |= value << s.u.decode.bsBits++;
      }
      else if (s.u.decode.bsBytes < s.u.decode.maxFrame)
      {
         nfc_push_byte(mem, s, s.u.decode.bsData);
         if (!nfca_parity(s.u.decode.bsData, value))
            s.u.decode.bsFlags |= NFC_FLAG_PARITY;
         s.u.decode.bsData = 0;
         s.u.decode.bsBits = 0;
      }
      else
      {
         nfca_reset(c, s, mem);
         return;
      }
   }

   s.u.decode.bsPrevious = pattern;
}

/* ---- listen SOF, 106k OOK subcarrier, NfcA.cpp:939-1090 ---- */
/* The pulse tracker of the listen-frame start (NfcA.cpp:976-1060): a rising correlation above the threshold opens a
 * pulse, its largest value half a symbol on marks the start, the falling one its end; true when a pulse of one symbol's
 * width has just ended (the start of frame, pattern D: the caller sets the symbol up from the record), false while it is
 * still looking - then the record is all it has touched. On its own because the wave decoder applies it in place
 * (nfc_wave_fast.hpp) where the window between guard and waiting time is open and the modulation is not too deep. */
NFC_DEV bool nfca_listen_ask_track(NfcMod &m, const NfcRate &rt, uint32_t clock, float s0)
{
   if (!m.symStart)
   {
      if (s0 > m.thr && s0 > m.peak)
      {
         m.peak = s0;
         m.peakTime = clock;
         m.winEnd = clock + rt.p4;
      }
   }
   else if (s0 < -m.thr && s0 < m.peak)
   {
      m.peak = s0;
      m.peakTime = clock;
   }

   if (clock != m.winEnd)
      return false;

   if (!m.symStart)
   {
      m.sync = m.peakTime + rt.p2;
      m.winEnd = m.winEnd + rt.p2;
      m.symStart = m.peakTime - rt.p2;
      m.peakTime = 0;
      m.peak = 0;
      return false;
   }

   m.symEnd = m.peakTime;
   m.pulses = m.symEnd - m.symStart;

   const uint32_t minimumWidth = rt.p1 - rt.p8;
   const uint32_t maximumWidth = rt.p1 + rt.p8;

   if (m.peakTime == 0 || m.pulses < minimumWidth || m.pulses > maximumWidth)
   {
      m.symStart = 0; m.symEnd = 0; m.sync = 0; m.winStart = 0; m.winEnd = 0; m.pulses = 0;
      m.peakTime = 0; m.peak = 0; m.auxTime = 0; m.aux = 0;
      return false;
   }

   return true;
}

NFC_DEV uint32_t nfca_listen_ask_start(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   /* this stage only forms S0 (NfcA.cpp:962-975): same ring, no S1 */
   const uint32_t cur = s.clock - rt.delay;
   const uint32_t pos = nfc_lock_pos(s);

   const float v = taps.f0;
   const float old = taps.pp;
   const float c2 = taps.c2;
   const float guardDev = taps.m0;
   const float deep = now.depth;
   const float sq = v * v * 10.0f;

   NFC_AT(mem, NFC_R_PROD, cur & NFC_PMASK) = sq;
   m.acc += sq;
   m.acc -= old;

   NFC_AT(mem, NFC_R_CORR, s.u.decode.lockBase + pos) = m.acc;
   float s0 = m.acc - c2;

   if (s.clock < s.u.decode.guardEnd)
      return SYM_NONE;

   if (s.clock == s.u.decode.guardEnd)
      m.thr = guardDev * (float)rt.p8;

   if (s.clock > s.u.decode.waitingEnd)
      return SYM_TIMEOUT;

   if (deep > c.minDepth[0])
      return SYM_TIMEOUT;

   if (!nfca_listen_ask_track(m, rt, s.clock, s0))
      return SYM_NONE;

   m.sync = m.symEnd + rt.p1;
   m.winStart = m.sync - rt.p8;
   m.winEnd = m.sync + rt.p8;
   m.thr = nfc_abs(m.peak * 0.25f);
   m.c0 = 0;
   m.c1 = 0;
   m.peakTime = 0;
   m.peak = 0;

   s.u.decode.symValue = 1;
   s.u.decode.symStart = m.symStart - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.delay;
   s.u.decode.symPattern = A_D;

   return A_D;
}

/* ---- listen symbols, 106k Manchester, NfcA.cpp:1095-1214 ---- */
NFC_DEV uint32_t nfca_listen_ask_symbol(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   NfcCorr k = nfc_correlate_power(mem, s.clock, m, rt, s.u.decode.lockBase, nfc_lock_pos(s), taps.f0, taps.pp, taps.c2, taps.c3);
   float sd = nfc_abs(k.s0 - k.s1);

   if (s.clock < m.winStart)
      return SYM_NONE;

   if (sd > m.peak)
   {
      m.peak = sd;
      m.peakTime = s.clock;
   }

   if (s.clock == m.sync)
   {
      m.cD = sd;
      m.c0 = k.s0;
      m.c1 = k.s1;
   }

   if (s.clock != m.winEnd)
      return SYM_NONE;

   if (m.cD > m.thr)
   {
      m.symStart = m.symEnd;
      m.symEnd = m.peakTime;
      m.thr = m.peak * 0.25f;

      if (m.c0 > m.c1)
      {
         m.symRise = m.sync;
         s.u.decode.symValue = 0;
         s.u.decode.symPattern = A_E;
      }
      else
      {
         m.symRise = m.sync - rt.p2;
         s.u.decode.symValue = 1;
         s.u.decode.symPattern = A_D;
      }
   }
   else
   {
      m.symStart = m.symEnd;
      m.symEnd = m.sync;
      m.symRise = 0;
      s.u.decode.symPattern = A_F;
   }

   m.sync = m.symEnd + rt.p1;
   m.winStart = m.sync - rt.p8;
   m.winEnd = m.sync + rt.p8;
   m.peakTime = 0;
   m.peak = 0;

   s.u.decode.symStart = m.symStart - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.delay;
   s.u.decode.symEdge = m.symRise - rt.delay;

   return s.u.decode.symPattern;
}

/* ---- listen SOF, BPSK (212k/424k), NfcA.cpp:1220-1329 ---- */
NFC_DEV uint32_t nfca_listen_bpsk_start(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now, const NfcDecTaps &taps)
{
   const NfcRate &rt = s.u.decode.rt;
   NfcMod &m = s.u.decode.lock;

   const float deep = now.depth;
   const float guardDev = taps.m0;
   const NfcPhase p = nfc_phase_product(mem, s.clock, rt, taps.f0, taps.f1, taps.pp);

   if (s.clock < s.u.decode.guardEnd)
      return SYM_NONE;

   if (s.clock == s.u.decode.guardEnd)
      m.thr = guardDev;

   if (s.clock > s.u.decode.waitingEnd)
      return SYM_TIMEOUT;

   if (deep > c.minDepth[0])
      return SYM_TIMEOUT;

   nfc_phase_integrate(m, p);

   if (m.phaseAcc > m.thr)
   {
      if (!m.symStart)
         m.symStart = s.clock;

      m.winEnd = s.clock + rt.p2;
   }

   if (!m.symEnd && (m.phaseAcc < 0 || s.clock == m.winEnd))
   {
      int preamble = (int)(s.clock - m.symStart);

      if (preamble < c.etu * 3 || preamble > c.etu * 4)
      {
         m.symStart = 0;
         m.symEnd = 0;
         m.winEnd = 0;
         return SYM_NONE;
      }

      m.symEnd = m.winEnd + rt.p2;
   }

   if (s.clock != m.winEnd)
      return SYM_NONE;

   m.sync = m.symEnd + rt.p2;
   m.lastPhase = m.phaseAcc;
   m.phaseThr = nfc_abs(m.phaseAcc * 0.25f);
   m.auxTime = 0;

   s.u.decode.symValue = 0;
   s.u.decode.symStart = m.symStart - rt.p1 - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.p1 - rt.delay;
   s.u.decode.symPattern = A_S;

   return A_S;
}

/* ---- listen symbols, BPSK, NfcA.cpp:1334-1421 ---- */
NFC_DEV uint32_t nfca_listen_bpsk_symbol(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcDecTaps &taps)
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
      return A_O;

   m.symStart = m.symEnd;
   m.symEnd = m.sync + rt.p2;
   m.sync = m.sync + rt.p1;
   m.lastPhase = m.phaseAcc;
   m.auxTime = 0;

   if (m.phaseAcc < -m.phaseThr)
   {
      s.u.decode.symValue = !s.u.decode.symValue;
      s.u.decode.symPattern = (s.u.decode.symPattern == A_M) ? A_N : A_M;
   }
   else
   {
      m.phaseThr = m.phaseAcc * 0.25f;
   }

   s.u.decode.symStart = m.symStart - rt.p1 - rt.delay;
   s.u.decode.symEnd = m.symEnd - rt.p1 - rt.delay;

   return s.u.decode.symPattern;
}

/* a listen frame is complete: it is classified, emitted and followed by the fall back to search at the end of
 * the decode step (nfc_finish_frame) */
NFC_DEV void nfca_finish_listen(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, uint32_t flags)
{
   nfc_pend_frame(s, NFC_FRAME_LISTEN, flags);
}

/* ---- one sample in locked NFC-A mode: decodeFrame, NfcA.cpp:416-803 ---- */
NFC_DEV void nfca_decode(const NfcConfig &c, NfcStreamState &s, const NfcLaneMem &mem, const NfcNow &now, const NfcDecTaps &taps)
{

   if (s.u.decode.frameType == NFC_FRAME_POLL)
   {
      uint32_t pattern = nfca_poll_symbol(c, s, mem, taps);

      if (pattern > SYM_TIMEOUT)
         nfca_poll_frame(c, s, mem, pattern);

      return;
   }

   if (s.u.decode.frameType != NFC_FRAME_LISTEN)
      return;

   if (s.u.decode.lockRate == 0)
   {
      if (!s.u.decode.frameStart)
      {
         uint32_t pattern = nfca_listen_ask_start(c, s, mem, now, taps);

         if (pattern == A_D || pattern == SYM_TIMEOUT)
            nfc_wait_ended(mem, s, 0u, pattern == SYM_TIMEOUT && s.clock > s.u.decode.waitingEnd);

         if (pattern == A_D)
            s.u.decode.frameStart = s.u.decode.symStart;
         else if (pattern == SYM_TIMEOUT)
            nfca_reset(c, s, mem);

         return;
      }

      uint32_t pattern = nfca_listen_ask_symbol(c, s, mem, taps);

      if (pattern <= SYM_TIMEOUT)
         return;

      bool frameEnd = false, truncated = false;

      if (pattern == A_F)
         frameEnd = true;
      else if (s.u.decode.bsBytes == s.u.decode.maxFrame)
         truncated = true;

      if (frameEnd || truncated)
      {
         if (s.u.decode.bsBytes > 0 || s.u.decode.bsBits == 4)
         {
            if (s.u.decode.bsBits == 4)
               nfc_push_byte(mem, s, s.u.decode.bsData);

            uint32_t flags = 0;
            if (s.u.decode.bsFlags & NFC_FLAG_PARITY)
               flags |= NFC_FLAG_PARITY;
            if (truncated)
               flags |= NFC_FLAG_TRUNCATED;
            if (s.u.decode.bsBytes == 1 && s.u.decode.bsBits == 4)
               flags |= NFC_FLAG_SHORT;

            nfca_finish_listen(c, s, mem, flags);
            return;
         }

         nfca_reset_search(s, s.u.decode.lock);
         return;
      }

      if (s.u.decode.symEdge)
         s.u.decode.frameEnd = s.u.decode.symEdge;

      if (s.u.decode.bsBits < 8)
      {
         s.u.decode.bsData |= (s.u.decode.symValue << s.u.decode.bsBits++);
      }
      else if (s.u.decode.bsBytes < s.u.decode.maxFrame)
      {
         nfc_push_byte(mem, s, s.u.decode.bsData);
         if (!nfca_parity(s.u.decode.bsData, s.u.decode.symValue))
            s.u.decode.bsFlags |= NFC_FLAG_PARITY;
         s.u.decode.bsData = 0;
         s.u.decode.bsBits = 0;
      }
      else
      {
         nfca_reset(c, s, mem);
      }

      return;
   }

   /* 212k / 424k: BPSK */
   if (!s.u.decode.frameStart)
   {
      uint32_t pattern = nfca_listen_bpsk_start(c, s, mem, now, taps);

      if (pattern == A_S || pattern == SYM_TIMEOUT)
         nfc_wait_ended(mem, s, 0u, pattern == SYM_TIMEOUT && s.clock > s.u.decode.waitingEnd);

      if (pattern == A_S)
         s.u.decode.frameStart = s.u.decode.symStart;
      else if (pattern == SYM_TIMEOUT)
         nfca_reset(c, s, mem);

      return;
   }

   uint32_t pattern = nfca_listen_bpsk_symbol(c, s, mem, taps);

   if (pattern <= SYM_TIMEOUT)
      return;

   bool frameEnd = false, truncated = false;

   if (pattern == A_O)
      frameEnd = true;
   else if (s.u.decode.bsBytes == s.u.decode.maxFrame)
      truncated = true;

   if (frameEnd || truncated)
   {
      if (s.u.decode.bsBits == 9)
      {
         nfc_push_byte(mem, s, s.u.decode.bsData);

         if (nfca_parity(s.u.decode.bsData, s.u.decode.bsParity))
            s.u.decode.bsFlags |= NFC_FLAG_PARITY;
      }

      if (s.u.decode.bsBytes > 0)
      {
         s.u.decode.frameEnd = s.u.decode.symEnd;

         uint32_t flags = 0;
         if (s.u.decode.bsFlags & NFC_FLAG_PARITY)
            flags |= NFC_FLAG_PARITY;
         if (truncated)
            flags |= NFC_FLAG_TRUNCATED;

         nfca_finish_listen(c, s, mem, flags);
         return;
      }

      nfca_reset(c, s, mem);
      return;
   }

   if (s.u.decode.bsBits < 8)
   {
      s.u.decode.bsData |= (s.u.decode.symValue << s.u.decode.bsBits);
   }
   else if (s.u.decode.bsBits < 9)
   {
      s.u.decode.bsParity = s.u.decode.symValue;
   }
   else
   {
      nfc_push_byte(mem, s, s.u.decode.bsData);
      if (!nfca_parity(s.u.decode.bsData, s.u.decode.bsParity))
         s.u.decode.bsFlags |= NFC_FLAG_PARITY;
      s.u.decode.bsData = s.u.decode.symValue;
      s.u.decode.bsBits = 0;
   }

   s.u.decode.bsBits++;
}

#endif
