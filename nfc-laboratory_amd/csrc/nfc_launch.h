/*
 * nfc_launch.h — launch descriptors shared by nfc_kernels.hip and the host runtime (nfcgpu.hip).
 */
#ifndef NFC_AMD_LAUNCH_H
#define NFC_AMD_LAUNCH_H

#include <stdint.h>

#include "nfc_types.h"

/* input of one stream slot for one launch */
struct NfcWork
{
   const uint8_t *data; /* device pointer: count*stride floats */
   uint32_t count;      /* samples */
   uint32_t stride;     /* 1 magnitude, 2 interleaved IQ (host bookkeeping; a launch has one format, NfcLaunch::uniformStride) */
   const uint32_t *tiles; /* windowed launches: tile flag words (nfc_scan.h) from the lane's first sample on */
};

struct NfcLaunch
{
   NfcStreamState *states; /* [maxStreams] register-resident part */
   NfcStreamCold *cold;    /* [maxStreams] protocol timing, touched at frame boundaries */
   float *rings;           /* [blocks][ringBlockFloats] */
   uint8_t *bytes;         /* [maxStreams][NFC_STREAM_BYTES] */
   uint32_t *sink;         /* packed frame records */
   uint32_t *sinkCtl;      /* [0] cursor (words), [1] dropped frames */
   const NfcWork *works;   /* per slot table, or null for the uniform layout below */
   const uint8_t *uniformBase;
   uint64_t uniformPitch;
   uint32_t uniformCount;
   uint32_t uniformStride; /* floats per sample of every row of the launch (both layouts): 1 magnitude, 2 IQ */
   uint32_t sinkWords;
   uint32_t firstBlock;
   uint32_t firstSlot;
   uint32_t slotCount;
   uint32_t ringBlockFloats;
   uint32_t warmFront;  /* windowed launches: leading samples of every lane that only run the front end ... */
   uint32_t warmCorr;   /* ... then samples that also keep the search correlators up, before the decoder goes live */
   struct NfcWindow *windows; /* windowed launches: per-slot records (stop / retired are written back) */
   const struct NfcScanJob *jobs; /* windowed launches: the submission's streams */
   uint32_t *laneStats; /* windowed launches: [0] samples stepped one by one by all lanes (in tiles' worth), [1] most by one lane, [2] lanes run,
                           [6] tiles the lanes took (what a pass decodes) */
   uint32_t launchSeq;  /* non-zero, distinct for every demodulation launch of a context (see NfcStreamState::served) */
   uint32_t forceExact; /* the host launches only the exact-modulo kernel: it takes every block, whatever the clocks say */
};

#endif
