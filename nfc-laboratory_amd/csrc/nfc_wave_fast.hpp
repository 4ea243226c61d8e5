/*
 * nfc_wave_fast.hpp — the wave decoder's bulk paths: for the mode the decoder is in, how many of the tile's remaining
 * samples change nothing but running sums and ring entries (nfc_wave.hpp), and the commit of that run.
 * Included by nfc_wave.hpp.
 */
#ifndef NFC_AMD_WAVE_FAST_HPP
#define NFC_AMD_WAVE_FAST_HPP

struct NfcWaveFast
{
   uint32_t unused;
};

NFC_DEV void nfc_wave_fast_begin(NfcWaveFast &f)
{
   f.unused = 0;
}

/* Samples from u.at on (at most up to n) that are committed in bulk; u.at is advanced past them. 0: the sample at u.at
 * has to be stepped. Called by every lane. */
NFC_DEV uint32_t nfc_wave_fast(const NfcConfig &c, NfcWaveUni &u, const NfcLaneMem &mem, NFC_WAVE_LDS NfcWaveLds *lds, NfcWaveFast &f, const NfcWaveTile &tile,
                               uint32_t n, bool upkeep, const NfcWaveItem &it)
{
   return 0u;
}

#endif
