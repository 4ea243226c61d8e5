/*
 * nfc_scan_launch.h — argument block of the kernels of the time-parallel path (nfc_scan.h), shared by
 * nfc_kernels.hip and the host runtime.
 */
#ifndef NFC_AMD_SCAN_LAUNCH_H
#define NFC_AMD_SCAN_LAUNCH_H

#include "nfc_launch.h"
#include "nfc_scan.h"

#ifndef NFC_AMD_SCAN_HPP
struct NfcScanParams
{
   float rangeK;
   float edgeK;
   float deepK;
   uint32_t chunkSamples;
   uint32_t warmSamples;
   uint32_t soloSamples;
   uint32_t offGridAlone; /* a stream with samples off the capture grid is decoded by its carry lane alone (the wave decoder walks
                             the running sums in the step's order there: nfc_wave_fast.hpp); 0: it takes the sequential kernels */
};
#endif

struct NfcScanArgs
{
   NfcScanJob *jobs;
   uint32_t nJobs;
   const NfcScanChunk *chunks;
   uint32_t nChunks;
   const NfcScanChunk *chunksMore; /* the scan kernel's list goes on here (a round's two repair lists in one launch) */
   uint32_t nChunksMore;
   uint32_t stride;            /* floats per sample of every job: 1 magnitude, 2 IQ */
   NfcScanParams params;
   const NfcStreamState *states; /* the streams' own slots (state a submission starts from) */
   NfcScanPoint *points;
   NfcScanSeam *seams;         /* [nChunks] */
   uint32_t *chunkEdge;        /* [nChunks] */
   NfcScanTile *tileStats;     /* per tile: what the walk recorded */
   uint32_t *tiles;            /* per tile: flag word (nfc_tile_flags, then NFC_TILE_RETIRE_OK) */
   NfcWindow *windows;         /* [firstWindowSlot + windowRoom]: entry i describes lane i of the lane arrays */
   NfcWork *works;             /* same indexing */
   uint32_t finalLaneSlot;     /* carry lanes occupy [0, nJobs), the lanes that regenerate a job's final state [finalLaneSlot, + nJobs) */
   uint32_t firstWindowSlot;   /* speculative lanes start here (multiple of 64) */
   uint32_t windowRoom;        /* speculative lanes there is room for */
   uint32_t *windowCount;      /* speculative lanes in use (device counter) */
   uint32_t *rerunCount;       /* jobs that need another decode pass (device counter) */
   NfcScanChunk *repairs;      /* chunks to walk again (nfc_seams_check), at most one per job and round */
   uint32_t *repairCount;
   NfcScanChunk *repairsEnv;   /* ... those of them whose envelope tracker alone is walked again (NFC_CHUNK_ENVELOPE), when they are listed apart (null: with the others) */
   uint32_t *repairEnvCount;
   uint32_t followChains;   /* nfc_envelope_kernel: a walk goes on into the chunks that inherit its chunk's envelope (nfc_envelope.hpp) */
   uint32_t *runList;          /* speculative lanes to run in the coming pass (indices into the lane arrays) */
   uint32_t *runCount;         /* entries of runList (device counter) */
   uint32_t *runNext;          /* next entry a persistent wave takes (device counter) */
   /* save area: a speculative lane that runs to the end of the submission (and is not the closing window) leaves a copy
    * of its rings and frame-assembly bytes here; should it be the stream's last lane, the stream's state is complete
    * without running it again (the persistent waves reuse a finished lane's ring storage) */
   float *saveRings;           /* [saveRoom][ringBlockFloats / 64] */
   uint8_t *saveBytes;         /* [saveRoom][NFC_STREAM_BYTES] */
   uint32_t *saveNext;         /* slots handed out (device counter) */
   uint32_t saveRoom;
   /* front-end planes for the wave decoder (nfc_wave.hpp): per sample {filtered, envelope, deviation, average} as the
    * decoder's front end leaves them after that sample, written by a second walk of the scan kernel from the verified
    * chunk starts (null: not written). Sample i of a job is record 64 * job.firstTile + i. */
   float *planes;
   uint32_t *planesStale;      /* [chunks] or null: set for a chunk whose start state is rewritten while the walk that writes the planes may
                                  already have read it (the walk over all chunks runs beside the later rounds of second walks: nfcgpu.hip);
                                  those chunks' planes are written again when the rounds are over */
   uint32_t planesPerChunk;    /* ... lanes per entry of the walk's list: entry e is a chunk, lane e * planesPerChunk + i walks its piece i of planesPiece
                                  samples (0: an entry is a lane's own piece or chunk) */
   uint32_t planesPiece;       /* the walk that writes them: samples per lane when it goes by the stored points (a multiple of NFC_SCAN_POINT: lane i
                                  of a job walks [i * planesPiece, ...) from the point stored there); 0: a lane per chunk, from the chunk's start */
};

#endif
