/*
 * nfc_wave_lone.hip - the wave decoder (nfc_wave.hip, the same text) compiled a second time for launches of few lanes of
 * work: a register budget of one wave per SIMD (256 architectural + 256 accumulation registers), so that nothing of the tile
 * loop lives in scratch memory. nfc_wave_kernel is compiled for three waves per SIMD (168 registers, 548 B of scratch per
 * lane of the wave): the right trade while there are thousands of lanes to keep resident, the wrong one for the later passes
 * of a submission, a short capture or the buffers of one receiver - a handful of wavefronts whose every scratch access is a
 * trip to memory on the critical path of the only wave of its SIMD. The host picks by the number of lanes in the launch
 * (nfcgpu.hip: NFCGPU_LONE_LANES).
 */
#define NFC_WAVE_KERNEL_NAME nfc_wave_lone_kernel
#define NFC_WAVE_WAVES 1
#include "nfc_wave.hip"
