/*
 * nfcgpu.hip — host runtime behind the C ABI of include/nfcgpu.h.
 *
 * Owns, per context: one HIP stream, the HBM-resident stream slots (state records, history rings,
 * frame assembly buffers), the device frame sink and the per-stream host frame queues. Streams are
 * grouped into stream blocks of 64 (one wavefront). There is deliberately no CPU decoding path here:
 * if HIP or the gfx950 code object is unavailable every entry point fails with NFCGPU_ENODEV/EHIP.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "../../include/nfcgpu.h"
#include "nfc_config.hpp"
#include "nfc_launch.h"
#include "nfc_scan_launch.h"

__global__ void nfc_demod_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L);
__global__ void nfc_demod_exact_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L);
__global__ void nfc_magnitude_kernel(const float2 *__restrict__ iq, float *__restrict__ out, uint64_t n);
__global__ void nfc_resample_radio_kernel(const float *__restrict__ in, uint64_t pitchFloats, uint32_t nBuffers, uint32_t n,
                                          float *__restrict__ out, uint64_t outPitchFloats, uint32_t capacityPairs,
                                          uint32_t *__restrict__ counts);
__global__ void nfc_demod_fixed_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L);
__global__ void nfc_demod_fixed_exact_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L);

/* the table the specialised kernels were compiled with (generated at build time, see gen_fixed_config.cpp) */
#define NFC_FIXED_FN static inline
#include "nfc_config_fixed.inc"
__global__ void nfc_init_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, uint32_t keepFrontEnd);
__global__ void nfc_scan_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A);
__global__ void nfc_seams_kernel(NfcScanArgs A, uint32_t first);
__global__ void nfc_tiles_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, uint32_t nTilesTotal);
__global__ void nfc_windows_kernel(NfcScanArgs A);
__global__ void nfc_carry_lanes_kernel(NfcScanArgs A, NfcLaunch real, NfcLaunch lanes, uint32_t pass);
__global__ void nfc_window_lanes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, NfcLaunch lanes, uint32_t pass, uint32_t order, uint32_t lenLo, uint32_t lenHi);
__global__ void nfc_final_lanes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A, NfcLaunch lanes);
__global__ void nfc_chain_kernel(NfcScanArgs A, NfcLaunch lanes, uint32_t maxPasses);
__global__ void nfc_finish_kernel(NfcScanArgs A, NfcLaunch real, NfcLaunch lanes);
__global__ void nfc_read_kernel(const float4 *__restrict__ data, uint64_t n, float *__restrict__ out);
__global__ void nfc_scan_planes_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A);
__global__ void nfc_envelope_kernel(const NfcConfig *__restrict__ cfgPtr, NfcScanArgs A);
__global__ void nfc_wave_kernel(const NfcConfig *__restrict__ cfgPtr, NfcLaunch L, NfcScanArgs A, uint32_t mode);
__global__ void nfc_planes_stale_kernel(NfcScanArgs A, const NfcScanChunk *all, uint32_t nAll, NfcScanChunk *out, uint32_t *count);

namespace {

constexpr uint32_t kMaxConfigs = 256; /* distinct decoder configurations in use at once (1.5 KB each on the device) */
constexpr uint32_t kRingBlockFloats = (4 * NFC_HIST_STORED + NFC_PROD + NFC_CORR_MAX) * NFC_LANES;

struct StreamInfo
{
   bool open = false;
   bool initialized = false; /* device state valid */
   bool needInit = true;
   bool explicitInit = false; /* the pending initialisation was asked for (initialize()), not caused by a buffer */
   bool listed = false; /* part of the batch being submitted */
   nfcgpu_params params {};
   float powerAtInit = 0.01f; /* carrier thresholds are derived when the decoder (re)initialises */
   uint32_t config = 0;
   bool hasConfig = false; /* `config` has been resolved (a stream opened but never fed refers to no table entry) */
   uint32_t derivedRate = 0; /* sample rate the running configuration was derived from (params.sample_rate is the stored one) */
   uint32_t clock = 0xFFFFFFFFu; /* mirror of the device sample clock (NfcStreamState::clock) */
   std::deque<nfcgpu_frame> queue;
};

/* true when a stream whose clock mirror reads `clock` is, during a submission of `count` samples, within 1024 samples
 * of its start or of the 32-bit clock wrap (same test as nfc_exact_span in nfc_kernels.hip, which decides per stream
 * block). The mirror itself is only advanced once the launch has been issued (commit_clock). */
bool exact_zone(uint32_t clock, uint32_t count)
{
   const uint32_t start = clock + 1u + 1024u;
   const uint32_t untilWrap = 0u - start;
   return count != 0 && (start < 2048u || untilWrap < count);
}

void commit_clock(StreamInfo &si, uint32_t count)
{
   si.clock += count;
}

struct ProfiledLaunch
{
   hipEvent_t start;
   hipEvent_t stop;
};

}

struct nfcgpu_ctx
{
   int device = 0;
   hipStream_t stream = nullptr;
   hipStream_t side = nullptr;       /* the carry lanes of a windowed pass run beside the speculative ones */
   uint32_t sideMode = 2;            /* NFCGPU_SIDE_STREAM: 0 one stream, 1 two fixed events, 2 events per pass */
   hipEvent_t forkEvent = nullptr, joinEvent = nullptr;
   hipStream_t low = nullptr;        /* lowest priority: the walk that writes the planes beside the rounds of second walks */
   uint32_t maxStreams = 0;
   uint32_t blocks = 0;

   NfcStreamState *dStates = nullptr;
   NfcStreamCold *dCold = nullptr;
   float *dRings = nullptr;
   uint8_t *dBytes = nullptr;
   uint32_t *dSink = nullptr;
   uint32_t *dSinkCtl = nullptr;
   uint64_t sinkWords = 0;
   uint32_t *ownSink = nullptr; /* the context's own sink, kept while a caller-provided one is attached */
   uint32_t *ownSinkCtl = nullptr;
   uint64_t ownSinkWords = 0;
   NfcWork *dWorks = nullptr;
   NfcConfig *dConfigs = nullptr;
   bool genericOnly = false; /* NFCGPU_GENERIC_KERNELS=1: never use the sample-rate-specialised kernels (testing) */
   /* Staging of host-resident input: two slots, each a device buffer with a pinned mirror. Caller memory is copied into
    * the mirror by the CPU and never handed to the GPU runtime (no page locking of memory whose lifetime belongs to the
    * caller); the copy to the device and the launches that read it are asynchronous, and an event recorded behind them
    * says when the slot may be written again. A submission therefore does not wait for its own kernels. */
   struct StageSlot
   {
      uint8_t *d = nullptr;
      uint8_t *h = nullptr;
      size_t bytes = 0;
      hipEvent_t done = nullptr;
      bool busy = false;
   };
   StageSlot stage[2];
   uint32_t stageNext = 0;
   bool inflight = false; /* something has been enqueued since the last stream synchronisation */

   std::vector<NfcConfig> configs;
   std::vector<StreamInfo> streams;
   std::vector<NfcWork> hWorks;
   std::vector<uint32_t> hSink;

   bool hold = false;
   bool profile = false;
   bool dirty = false; /* work submitted since last sync */
   uint32_t launchSeq = 0; /* stamp of the last demodulation launch (NfcLaunch::launchSeq) */

   /* ---- time-parallel path (nfc_scan.h): device buffers, grown on demand and kept ---- */
   bool windowed = true;           /* NFCGPU_WINDOWED=0 switches the path off */
   uint32_t windowedMinSamples = 32768; /* shortest submission (per stream) worth cutting into windows */
   uint32_t scanChunk = 8192;      /* samples per scan chunk, at least: short chunks = many lanes (the walk is latency-bound per wave) */
   uint32_t scanLanes = 32768;     /* chunks a large submission is cut into, at least (NFCGPU_SCAN_LANES) */
   bool scanChunkFixed = false;    /* NFCGPU_SCAN_CHUNK given: no sizing by the submission */
   uint32_t blockSamples = 1u << 23; /* a few long busy streams are decoded this many samples at a time (NFCGPU_BLOCK_SAMPLES) */
   bool inBlocks = false;
   uint32_t scanWarm = 4096;       /* samples walked ahead of a chunk (round 4: 6144 -> 4096; config 5 dense: scan and second walks 94 -> 79 ms per step, as many chunks walked again) */
   uint32_t maxPasses = 32;        /* decode passes before a stream of a large submission gives up (sequential path). Round 4: 12 -> 32: a late pass of a
                                      few lanes is 10-20 ms, the sequential kernels take seconds for a stream of 2^20 samples (256 dense streams x 2^20
                                      cut into 64 lanes each: one stream in 768 needed a thirteenth pass, and the step took 1035 instead of 305 ms) */
   uint32_t maxPassesFew = 48;     /* the same for submissions of fewer streams than a wave has lanes: the sequential path would crawl */
   struct DevBuf
   {
      void *ptr = nullptr;
      size_t bytes = 0;
   };
   DevBuf wRepairs, wJobs, wChunks, wPoints, wSeams, wChunkEdge, wTiles, wTileStats, wWindows, wWorks, wCounters, wRunList;
   uint32_t busyPercent = 8;   /* a stream with more than this share of busy tiles is "busy": few long busy streams are decoded in blocks */
   DevBuf vStates, vCold, vRings, vBytes, vSink, vSinkCtl, vSaveRings, vSaveBytes;
   uint32_t longFirst = 32768;      /* the run list of a pass takes its lanes longest first, by classes of their length down to this one (NFCGPU_LONG_FIRST; 0: as they come) */
   uint32_t lanesWanted = 4096;     /* lanes a large busy submission is cut into, at least (NFCGPU_LANES_WANTED; 0: always NFC_WINDOW_CUT apart). Round 4: 16384 -> 4096: longer lanes need fewer passes (512 dense streams x 2^20: 320 -> 273 ms per step; 4096 streams are at NFCGPU_CUT_MAX either way) */
   uint32_t cutMax = 1u << 19;      /* ... but never further apart than this (NFCGPU_CUT_MAX). Round 5: 2^19 instead of 2^17 - config 5 at 2^17, 2^18,
                                       2^19, 2^20: 458, 456, 452, 452 ms per step (three runs each at 2^17 and 2^19: +-1 ms); 2^16: 500, 2^15: 555. Most
                                       lanes begin after quiet signal, not at a cut; the fewer cuts, the fewer guesses */
   uint32_t stagingWords = 0;       /* NFCGPU_STAGING_WORDS: cap on the lanes' staging sink (0: none) */
   uint32_t soloSamples = 1u << 15; /* streams this short are decoded by their carry lane alone, in one pass (NFCGPU_SOLO_SAMPLES). Round 4: 2^18 -> 2^16,
                                       the lanes being what they now are: the bundled captures of 100 k - 200 k samples 25 / 39 / 49 -> 17 / 27 / 37 ms.
                                       Round 6: 2^16 -> 2^15. A 65536-sample buffer is what the reference's task hands the decoder per call
                                       (TS/main.cpp:163-165, RadioDecoderTask.cpp:377-401), and a lane without windows can retire to nobody: it walked
                                       all 1024 tiles of a buffer with nothing in it (7 ms). The task on the shim in its default mode, dense / sparse
                                       WAV: 4.9 / 6.4 -> 8.3 / 15.1 MS/s (profiles/r06/shim_default_mode.txt) */
   DevBuf wPlanes, wPlaneChunks;   /* front-end planes (NfcScanArgs::planes) and the chunk list of the walk that writes them */
   DevBuf wRepairsEnv;             /* the chunks of a round whose envelope tracker alone is walked again (nfc_envelope_kernel) */
   uint32_t envelopeMax = 16384;   /* ... when the round lists at most this many of them (NFCGPU_ENVELOPE_KERNEL; 0: never, one list for the scan kernel).
                                      Round 5: a wavefront per chunk (nfc_envelope.hpp). Round 4's kernel had a lane per chunk - good for the few
                                      4096-sample chunks of a short capture's rounds, the wrong shape for the long lists of a large submission
                                      (thousands of lanes reading 32768-sample chunks a cache line each: 79 -> 122 ms per step of the headline,
                                      profiles/r04/ab_envelope) - and was given lists of at most 64 */
   DevBuf wPlanesStale;            /* NfcScanArgs::planesStale */
   uint32_t planesBeside = 1;      /* the walk that writes a large submission's planes runs on the side stream beside the rounds of second walks that follow the
                                      first (NFCGPU_PLANES_BESIDE; 0: after them, as until round 5) */
   uint32_t planesBesidePiece = 2048; /* ... samples per lane of that walk, from the stored points (NFCGPU_PLANES_BESIDE_PIECE; 0: a lane per chunk, from its start).
                                      Config 5, ms per step: after the rounds 452.7; beside them a lane per chunk 442.8, per 8192 / 2048 / 512 samples 438.5 / 436.7 / 437.6 */
   uint32_t planesPiece = 512;     /* samples per lane of the walk that writes a small submission's front-end planes (NFCGPU_PLANES_PIECE; 0: a lane per chunk) */
   uint32_t envelopeFollowMax = 1024; /* ... and a walk goes on through the chain of chunks that inherit its chunk's envelope when the round lists at
                                         most this many (NFCGPU_ENVELOPE_FOLLOW): the tail of rounds with a few chunks each becomes one or two rounds
                                         (a short capture: ten rounds -> three, 3.9 -> 2.6 ms; config 5: seven -> five). Not for the long lists: a
                                         chain is then one wavefront's serial work while the rest of the device waits (seams 40 -> 46 ms with it) */
   std::vector<ProfiledLaunch> timedScan, timedWindow, timedWave, timedPlanes;
   hipEvent_t epoch = nullptr;      /* recorded when the statistics start over: the time base of the launch intervals below */
   std::vector<std::pair<float, float>> waveSpans; /* [start, stop) of every wave decoder launch since, ms after `epoch` */
   double waveBusyMs = 0.0;         /* ... and the union of those that have been folded away (fold_wave_spans) */

   /* ---- frame gather over RCCL (nfcgpu_comm_*) ---- */
   void *comm = nullptr;
   int commRank = 0, commRanks = 0;
   uint32_t *dCounts = nullptr;
   std::vector<ProfiledLaunch> timed;
   std::vector<hipEvent_t> eventPool;
   nfcgpu_stats stats {};
   std::string lastError;

};

namespace {

int fail(nfcgpu_ctx *ctx, int code, const char *what, hipError_t err = hipSuccess)
{
   if (ctx)
   {
      ctx->lastError = what;
      if (err != hipSuccess)
      {
         ctx->lastError += ": ";
         ctx->lastError += hipGetErrorString(err);
      }
   }
   return code;
}

#define HIP_TRY(ctx, call)                                   \
   do                                                        \
   {                                                         \
      hipError_t err__ = (call);                             \
      if (err__ != hipSuccess)                               \
         return fail((ctx), NFCGPU_EHIP, #call, err__);      \
   } while (0)

NfcLaunch base_launch(nfcgpu_ctx *ctx)
{
   NfcLaunch L;
   std::memset(&L, 0, sizeof(L));
   L.states = ctx->dStates;
   L.cold = ctx->dCold;
   L.rings = ctx->dRings;
   L.bytes = ctx->dBytes;
   L.sink = ctx->dSink;
   L.sinkCtl = ctx->dSinkCtl;
   L.sinkWords = (uint32_t)ctx->sinkWords;
   L.ringBlockFloats = kRingBlockFloats;
   return L;
}

/* true when every sample-rate-derived constant of `cfg` equals the table compiled into the specialised kernels
 * (thresholds and the enable mask are run-time values there as well) */
bool matches_fixed_table(const NfcConfig &cfg)
{
   if (cfg.sampleRate != NFC_FIXED_SAMPLE_RATE)
      return false;

   NfcConfig probe = cfg;
   nfc_fixed_config(probe);
   return std::memcmp(&probe, &cfg, sizeof(cfg)) == 0;
}

/* find or create the device-side NfcConfig for a stream's parameters */
int resolve_config(nfcgpu_ctx *ctx, StreamInfo &si)
{
   NfcHostParams hp;
   hp.sampleRate = si.derivedRate;
   hp.enabled = si.params.tech_mask & 0xF;
   hp.powerLevelThreshold = si.params.power_level_threshold;
   for (int t = 0; t < 4; t++)
   {
      hp.corrThreshold[t] = si.params.corr_threshold[t];
      hp.minDepth[t] = si.params.min_modulation_depth[t];
      hp.maxDepth[t] = si.params.max_modulation_depth[t];
   }

   NfcConfig cfg;
   if (!nfc_build_config(hp, cfg))
      return fail(ctx, NFCGPU_ERATE, "sample rate not decodable with the fixed history depth");

   /* signalLow/HighThreshold are only recomputed by initialize() (NfcDecoder.cpp:327-329) */
   cfg.lowThreshold = si.powerAtInit / 1.25f;
   cfg.highThreshold = si.powerAtInit * 1.25f;

   for (uint32_t i = 0; i < ctx->configs.size(); i++)
   {
      if (std::memcmp(&ctx->configs[i], &cfg, sizeof(cfg)) == 0)
      {
         si.config = i;
         si.hasConfig = true;
         return NFCGPU_OK;
      }
   }

   if (ctx->configs.size() < kMaxConfigs)
   {
      ctx->configs.push_back(cfg);
      si.config = (uint32_t)ctx->configs.size() - 1;
   }
   else
   {
      /* the table is full: take the place of a configuration no open stream refers to any more (parameters changed
       * since, streams closed). Launches that may still read it are waited for first. */
      std::vector<bool> used(kMaxConfigs, false);
      for (const StreamInfo &other: ctx->streams)
      {
         if (other.open && &other != &si && other.hasConfig)
            used[other.config] = true;
      }

      uint32_t slot = kMaxConfigs;
      for (uint32_t i = 0; i < kMaxConfigs && slot == kMaxConfigs; i++)
      {
         if (!used[i])
            slot = i;
      }

      if (slot == kMaxConfigs)
         return fail(ctx, NFCGPU_ENOMEM, "too many distinct decoder configurations in use at once");

      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      ctx->configs[slot] = cfg;
      si.config = slot;
   }

   si.hasConfig = true;

   HIP_TRY(ctx, hipMemcpyAsync(ctx->dConfigs + si.config, &cfg, sizeof(cfg), hipMemcpyHostToDevice, ctx->stream));
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); /* cfg is a stack object */

   return NFCGPU_OK;
}

/* The reference keeps two things apart: the sample rate it stores (setSampleRate(), or taken from a buffer whose rate
 * differs from the stored one, NfcDecoder.cpp:383-388) and the parameters initialize() derived from the rate stored at
 * that moment (NfcDecoder.cpp:295-360). A buffer only re-initialises the decoder when its rate differs from the stored
 * one, so after setSampleRate(x) buffers labelled x are decoded with the parameters of the previous rate. The one thing
 * not reproduced: parameters derived while the stored rate was still 0 (initialize() before any rate is known, then the
 * rate given through the setter) leave the reference with NaN filter weights and no output; here they are derived when
 * the first buffer arrives. */
int adopt_sample_rate(nfcgpu_ctx *ctx, StreamInfo &si, uint32_t sampleRate)
{
   if (sampleRate == 0)
      return fail(ctx, NFCGPU_EINVAL, "sample rate must be non-zero");

   if (si.params.sample_rate != sampleRate)
   {
      /* frames are stamped with the rate stored when they were produced (NfcDecoder.cpp frame.setSampleRate): collect
       * what the sink holds before the stored rate changes */
      if (si.params.sample_rate != 0 && ctx->dirty && !ctx->hold)
      {
         int rc = nfcgpu_sync(ctx);
         if (rc != NFCGPU_OK && rc != NFCGPU_EOVERFLOW)
            return rc;
      }

      si.params.sample_rate = sampleRate;
      si.derivedRate = sampleRate;
      si.needInit = true;
      si.explicitInit = false; /* the reference initialises again, with what is set now */
   }
   else if (!si.initialized || si.derivedRate == 0)
   {
      si.needInit = true;
   }

   if (si.needInit)
   {
      if (si.derivedRate == 0)
         si.derivedRate = sampleRate;

      /* the carrier thresholds follow the power level of the moment of the initialisation (NfcDecoder.cpp:327-329):
       * the moment of initialize() if that is what is pending, now otherwise */
      if (!si.explicitInit)
         si.powerAtInit = si.params.power_level_threshold;

      si.explicitInit = false;
      return resolve_config(ctx, si);
   }

   return NFCGPU_OK;
}

/* run nfc_init_kernel for every stream in [first, first+count) that needs it (listedOnly: and is part of the batch
 * being submitted; a stream opened but not yet fed has no configuration resolved); contiguous runs with equal
 * (config, keep) share one launch */
int initialize_pending(nfcgpu_ctx *ctx, uint32_t first, uint32_t count, bool listedOnly)
{
   uint32_t i = first;
   const uint32_t end = first + count;

   auto due = [&](const StreamInfo &si) { return si.open && si.needInit && (si.listed || !listedOnly); };

   while (i < end)
   {
      StreamInfo &si = ctx->streams[i];

      if (!due(si))
      {
         i++;
         continue;
      }

      const uint32_t cfg = si.config;
      const bool keep = si.initialized;
      uint32_t j = i;

      while (j < end && due(ctx->streams[j]) && ctx->streams[j].config == cfg && ctx->streams[j].initialized == keep)
         j++;

      NfcLaunch L = base_launch(ctx);
      L.firstSlot = i;
      L.slotCount = j - i;

      const uint32_t threads = 64;
      const uint32_t grid = (L.slotCount + threads - 1) / threads;

      hipLaunchKernelGGL(nfc_init_kernel, dim3(grid), dim3(threads), 0, ctx->stream, ctx->dConfigs + cfg, L, keep ? 1u : 0u);
      HIP_TRY(ctx, hipGetLastError());

      for (uint32_t k = i; k < j; k++)
      {
         ctx->streams[k].needInit = false;
         ctx->streams[k].initialized = true;
         ctx->streams[k].clock = 0xFFFFFFFFu;
      }

      i = j;
   }

   return NFCGPU_OK;
}

hipEvent_t take_event(nfcgpu_ctx *ctx)
{
   if (!ctx->eventPool.empty())
   {
      hipEvent_t e = ctx->eventPool.back();
      ctx->eventPool.pop_back();
      return e;
   }
   hipEvent_t e = nullptr;
   (void)hipEventCreate(&e);
   return e;
}

int launch_demod(nfcgpu_ctx *ctx, uint32_t config, NfcLaunch &L, uint64_t samples, bool exactPossible, bool exactOnly)
{
   const uint32_t firstBlock = L.firstSlot / NFC_LANES;
   const uint32_t lastBlock = (L.firstSlot + L.slotCount - 1) / NFC_LANES;

   L.firstBlock = firstBlock;

   ProfiledLaunch pl {nullptr, nullptr};

   if (ctx->profile)
   {
      pl.start = take_event(ctx);
      pl.stop = take_event(ctx);
      HIP_TRY(ctx, hipEventRecord(pl.start, ctx->stream));
   }

   const bool fixed = !ctx->genericOnly && matches_fixed_table(ctx->configs[config]);

   /* every stream block picks its kernel from the device state; a launch in which every stream needs the exact
    * variant (the first buffer of freshly opened streams) does not need the common kernel at all */
   L.forceExact = exactOnly ? 1u : 0u;

   if (++ctx->launchSeq == 0)
      ctx->launchSeq = 1;
   L.launchSeq = ctx->launchSeq;

   if (!exactOnly)
   {
      hipLaunchKernelGGL(fixed ? nfc_demod_fixed_kernel : nfc_demod_kernel, dim3(lastBlock - firstBlock + 1), dim3(NFC_LANES), 0,
                         ctx->stream, ctx->dConfigs + config, L);
      HIP_TRY(ctx, hipGetLastError());
   }

   /* stream blocks near their start / the clock wrap skip the kernel above and are handled by this one */
   if (exactPossible)
   {
      hipLaunchKernelGGL(fixed ? nfc_demod_fixed_exact_kernel : nfc_demod_exact_kernel, dim3(lastBlock - firstBlock + 1),
                         dim3(NFC_LANES), 0, ctx->stream, ctx->dConfigs + config, L);
      HIP_TRY(ctx, hipGetLastError());
   }

   if (ctx->profile)
   {
      HIP_TRY(ctx, hipEventRecord(pl.stop, ctx->stream));
      ctx->timed.push_back(pl);
   }

   ctx->stats.launches++;
   ctx->stats.samples += samples;
   ctx->dirty = true;

   return NFCGPU_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* time-parallel path (nfc_scan.h)                                                             */
/* ------------------------------------------------------------------------------------------ */

struct WindowedItem
{
   uint32_t slot;
   const uint8_t *data; /* device */
   uint32_t count;
};

int grow(nfcgpu_ctx *ctx, nfcgpu_ctx::DevBuf &b, size_t bytes)
{
   if (bytes <= b.bytes)
      return NFCGPU_OK;

#ifdef NFCGPU_EMULATED_TEST_BUILD
   /* (test build: a device that cannot give the front-end planes more than this - the fallbacks of run_windowed) */
   if (const char *limit = std::getenv("NFCGPU_TEST_ALLOC_LIMIT"))
   {
      if (&b == &ctx->wPlanes && bytes > std::strtoull(limit, nullptr, 10))
         return fail(ctx, NFCGPU_ENOMEM, "device allocation for the time-parallel path failed (test limit)");
   }
   if (const char *limit = std::getenv("NFCGPU_TEST_ALLOC_LIMIT_LANES"))
   {
      /* (the same for a buffer that is grown elsewhere in run_windowed: the lanes' decoder states) */
      if (&b == &ctx->vStates && bytes > std::strtoull(limit, nullptr, 10))
         return fail(ctx, NFCGPU_ENOMEM, "device allocation for the time-parallel path failed (test limit)");
   }
#endif

   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

   if (b.ptr)
      (void)hipFree(b.ptr);

   b.ptr = nullptr;
   b.bytes = 0;

   /* some room to grow into (a quarter; little for the buffers that are gigabytes), none when that does not fit */
   size_t want = bytes + (bytes < ((size_t)1 << 30) ? bytes / 4 : bytes / 32) + 256;

   if (hipMalloc(&b.ptr, want) != hipSuccess)
   {
      (void)hipGetLastError();
      want = bytes;

      if (hipMalloc(&b.ptr, want) != hipSuccess)
      {
         size_t freeBytes = 0, totalBytes = 0;
         (void)hipGetLastError();
         (void)hipMemGetInfo(&freeBytes, &totalBytes);
         b.ptr = nullptr;
         char what[160];
         std::snprintf(what, sizeof(what), "device allocation for the time-parallel path failed (%.2f GiB wanted, %.2f of %.2f GiB free)", (double)bytes / (double)(1 << 30),
                       (double)freeBytes / (double)(1 << 30), (double)totalBytes / (double)(1 << 30));
         return fail(ctx, NFCGPU_ENOMEM, what);
      }
   }

   b.bytes = want;
   return NFCGPU_OK;
}

/* the sequential kernels over a subset of slots (fallback of the time-parallel path) */
int launch_sequential(nfcgpu_ctx *ctx, uint32_t config, const std::vector<WindowedItem> &items, uint32_t stride)
{
   if (items.empty())
      return NFCGPU_OK;

   /* a long first buffer of fresh streams: only its first samples need the exact-modulo kernel (which is chosen per
    * launch and is the slower one), so they get a launch of their own */
   {
      const uint32_t head = 2048;
      bool split = false;
      for (const WindowedItem &it: items)
         split = split || (ctx->streams[it.slot].clock == 0xFFFFFFFFu && it.count > 2 * head);

      if (split)
      {
         std::vector<WindowedItem> first, rest;
         for (const WindowedItem &it: items)
         {
            const uint32_t n = it.count < head ? it.count : head;
            first.push_back(WindowedItem {it.slot, it.data, n});
            if (it.count > n)
               rest.push_back(WindowedItem {it.slot, it.data + (size_t)n * stride * 4, it.count - n});
         }

         int rc = launch_sequential(ctx, config, first, stride);
         if (rc)
            return rc;
         return launch_sequential(ctx, config, rest, stride);
      }
   }

   uint32_t first = 0xFFFFFFFFu, last = 0;
   uint64_t samples = 0;
   bool exactPossible = false, exactOnly = true;

   for (const WindowedItem &it: items)
   {
      first = it.slot < first ? it.slot : first;
      last = it.slot > last ? it.slot : last;
   }

   std::vector<NfcWork> table(last - first + 1);
   for (NfcWork &w: table)
   {
      w.data = nullptr;
      w.count = 0;
      w.stride = 1;
      w.tiles = nullptr;
   }

   for (const WindowedItem &it: items)
   {
      NfcWork &w = table[it.slot - first];
      w.data = it.data;
      w.count = it.count;
      w.stride = stride;
      samples += it.count;
      const bool exact = exact_zone(ctx->streams[it.slot].clock, it.count);
      exactPossible = exactPossible || exact;
      exactOnly = exactOnly && (exact || it.count == 0);
   }

   HIP_TRY(ctx, hipMemcpyAsync(ctx->dWorks + first, table.data(), sizeof(NfcWork) * table.size(), hipMemcpyHostToDevice, ctx->stream));
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); /* `table` is a local */

   NfcLaunch L = base_launch(ctx);
   L.works = ctx->dWorks;
   L.uniformStride = stride;
   L.firstSlot = first;
   L.slotCount = last - first + 1;

   int rc = launch_demod(ctx, config, L, samples, exactPossible, exactOnly);
   if (rc)
      return rc;

   for (const WindowedItem &it: items)
      commit_clock(ctx->streams[it.slot], it.count);

   return NFCGPU_OK;
}

/* may these streams take the time-parallel path for this submission? (one configuration, one sample format) */
bool windowed_eligible(nfcgpu_ctx *ctx, uint32_t config, const std::vector<WindowedItem> &items)
{
   if (!ctx->windowed || ctx->genericOnly || items.empty() || !matches_fixed_table(ctx->configs[config]))
      return false;

   for (const WindowedItem &it: items)
   {
      if (it.count < ctx->windowedMinSamples)
         return false;

      /* stay clear of the 32-bit wrap of the sample clock: the lanes number their rings from their own first sample
       * and never take the exact-modulo route (a fresh stream, clock 0xFFFFFFFF, is handled by its carry lane) */
      const uint32_t clock = ctx->streams[it.slot].clock;
      if (clock != 0xFFFFFFFFu && (uint64_t)clock + it.count + 4096u >= 0xFFFFFFFFull)
         return false;
   }

   return true;
}

void record_span(nfcgpu_ctx *ctx, std::vector<ProfiledLaunch> &into, ProfiledLaunch &pl, bool begin, hipStream_t on = nullptr)
{
   if (!ctx->profile)
      return;

   if (!on)
      on = ctx->stream;

   if (begin)
   {
      pl.start = take_event(ctx);
      pl.stop = take_event(ctx);
      (void)hipEventRecord(pl.start, on);
   }
   else
   {
      (void)hipEventRecord(pl.stop, on);
      into.push_back(pl);
   }
}

int run_windowed(nfcgpu_ctx *ctx, uint32_t config, const std::vector<WindowedItem> &items, uint32_t stride);
int launch_sequential(nfcgpu_ctx *ctx, uint32_t config, const std::vector<WindowedItem> &items, uint32_t stride);

/* the same submission, `blockSamples` at a time (0: the context's block length) */
int run_in_blocks(nfcgpu_ctx *ctx, uint32_t config, const std::vector<WindowedItem> &items, uint32_t stride, uint32_t blockSamples = 0)
{
   uint32_t longest = 0;
   for (const WindowedItem &it: items)
      longest = it.count > longest ? it.count : longest;

   if (!blockSamples)
      blockSamples = ctx->blockSamples;

   int rc = NFCGPU_OK;
   ctx->inBlocks = true;

   for (uint64_t at = 0; at < longest && rc == NFCGPU_OK; at += blockSamples)
   {
      std::vector<WindowedItem> block;

      for (const WindowedItem &it: items)
      {
         if (it.count > at)
         {
            const uint64_t left = it.count - at;
            block.push_back(WindowedItem {it.slot, it.data + (size_t)at * stride * 4, (uint32_t)(left < blockSamples ? left : blockSamples)});
         }
      }

      rc = windowed_eligible(ctx, config, block) ? run_windowed(ctx, config, block, stride) : launch_sequential(ctx, config, block, stride);
   }

   ctx->inBlocks = false;
   return rc;
}

/* One submission of `items` (all of configuration `config`, `stride` floats per sample, data resident on the device)
 * through scan -> windows -> windowed decode -> chain -> finish; streams the path cannot vouch for (samples off the
 * int16 grid, a seam that did not verify, no settled chain) are then decoded sequentially from their untouched state. */
int run_windowed(nfcgpu_ctx *ctx, uint32_t config, const std::vector<WindowedItem> &items, uint32_t stride)
{
   const auto entered = std::chrono::steady_clock::now(); /* (the stage log counts the host's tables from here) */
   const uint32_t nJobs = (uint32_t)items.size();
   const NfcConfig &cfg = ctx->configs[config];

   /* A work buffer the device cannot give (NFCGPU_ENOMEM from grow()) is not the end of a submission as long as nothing of the
    * streams has been touched - which holds up to the finish: the scan and the lanes only read the streams' state. The
    * submission is then decoded a quarter of its length at a time (a quarter of every work buffer), and if that does not fit
    * either by the sequential kernels, which need none. Any other error is the caller's. */
   auto withoutTheMemory = [&](int code) -> int {
      if (code != NFCGPU_ENOMEM)
         return code;

      (void)hipGetLastError();

      uint32_t longest = 0;
      for (const WindowedItem &it: items)
         longest = it.count > longest ? it.count : longest;

      const uint32_t quarter = longest / 4u / NFC_SCAN_POINT * NFC_SCAN_POINT;

      if (!ctx->inBlocks && quarter >= 65536u && quarter >= ctx->windowedMinSamples)
         return run_in_blocks(ctx, config, items, stride, quarter);

      ctx->stats.fallback_streams += nJobs;
      return launch_sequential(ctx, config, items, stride);
   };

   NfcScanParams sp;
   {
      float corr = 3.0e38f;
      if (cfg.enabled & 1u) corr = cfg.corrThreshold[0] < corr ? cfg.corrThreshold[0] : corr;
      if (cfg.enabled & 4u) corr = cfg.corrThreshold[2] < corr ? cfg.corrThreshold[2] : corr;
      if (cfg.enabled & 8u) corr = cfg.corrThreshold[3] < corr ? cfg.corrThreshold[3] : corr;
      sp.rangeK = corr < 1.0e30f ? 0.49f * corr : 3.0e38f;
      sp.edgeK = (cfg.enabled & 2u) ? 0.99f * cfg.minDepth[1] : 3.0e38f;
      float deep = 1.0f;
      for (int t = 0; t < 4; t++)
         if ((cfg.enabled >> t) & 1u)
            deep = cfg.maxDepth[t] < deep ? cfg.maxDepth[t] : deep;
      sp.deepK = 0.98f * deep;
      sp.chunkSamples = ctx->scanChunk;
      sp.warmSamples = ctx->scanWarm;
      sp.soloSamples = ctx->soloSamples;
      sp.offGridAlone = 1u;

      /* Every chunk pays the warm-up again, so chunks should be as long as the machine allows: one lane per chunk, and
       * 131072 lanes (256 CUs x 4 SIMDs x 2 waves of the scan kernel's 204 registers x 64) are resident at a time.
       * Measured on 4096 streams x 2^20 idle samples: 8192 -> 1347, 16384 -> 1673, 32768 -> 1906 GB/s. */
      if (!ctx->scanChunkFixed)
      {
         uint64_t total = 0;
         for (const WindowedItem &it: items)
            total += it.count;

         /* (round 4: a sixteenth of that is enough lanes. What a submission of 2^29 samples - 512 busy streams, an eighth of
          * config 5 - pays for are the rounds of second walks, a launch and a trip to the host each, and a chain of chunks that
          * inherit a wrong envelope from each other is as many rounds as it has chunks: 27 rounds of 4096-sample chunks, 8 of
          * 32768. 512 / 1024 dense streams x 2^20: 352 -> 321 ms per step.) */
         uint64_t chunk = total / ctx->scanLanes / NFC_SCAN_POINT * NFC_SCAN_POINT;
         if (chunk > 32768u)
            chunk = 32768u;
         if (chunk > sp.chunkSamples)
            sp.chunkSamples = (uint32_t)chunk;

         /* A small submission is one a caller waits for (a capture, a receiver's block): what counts is the time of the
          * longest walk, chunk + warm-up at ~0.4 us per sample and lane. Shorter chunks and a warm-up that just covers the
          * slowest recurrence (the average: 0.995^k) cut it; seams that do not verify cost a short second walk now. */
         if (total <= (4u << 20))
         {
            if (sp.chunkSamples > 4096u)
               sp.chunkSamples = 4096u;
            if (sp.warmSamples > 3072u)
               sp.warmSamples = 3072u;
         }
      }
   }

   /* job and chunk tables */
   std::vector<NfcScanJob> jobs(nJobs);
   std::vector<NfcScanChunk> chunks;
   uint32_t tiles = 0, points = 0, tilesMost = 0;
   uint64_t totalSamples = 0;

   for (uint32_t j = 0; j < nJobs; j++)
   {
      NfcScanJob &job = jobs[j];
      std::memset(&job, 0, sizeof(job));
      job.data = items[j].data;
      job.count = items[j].count;
      job.slot = items[j].slot;
      job.firstChunk = (uint32_t)chunks.size();
      job.chunks = (job.count + sp.chunkSamples - 1) / sp.chunkSamples;
      job.firstTile = tiles;
      job.firstPoint = points;
      tiles += (job.count + NFC_SCAN_TILE - 1) / NFC_SCAN_TILE;
      tilesMost = std::max(tilesMost, (job.count + NFC_SCAN_TILE - 1) / NFC_SCAN_TILE);
      points += job.count / NFC_SCAN_POINT + 1;
      totalSamples += job.count;

      for (uint32_t k = 0; k < job.chunks; k++)
         chunks.push_back(NfcScanChunk {j, k});
   }

   /* Lanes inside busy signal: NFC_WINDOW_CUT samples apart when the submission is small (every lane is parallelism),
    * further apart - fewer warm-ups, fewer hand-overs to go wrong - when that still leaves several lanes per wave slot
    * of the machine (NFCGPU_LANES_WANTED, default 16384 = 8 per slot of 256 CUs x 8 waves) */
   {
      uint64_t cut = ctx->lanesWanted ? totalSamples / ctx->lanesWanted : 0u;
      cut = cut / NFC_SCAN_POINT * NFC_SCAN_POINT;
      cut = cut < NFC_WINDOW_CUT ? NFC_WINDOW_CUT : (cut > ctx->cutMax ? ctx->cutMax : cut);

      for (NfcScanJob &job: jobs)
         job.cut = (uint32_t)cut;
   }

   const uint32_t nChunks = (uint32_t)chunks.size();
   const uint32_t finalLaneSlot = (nJobs + NFC_LANES - 1) / NFC_LANES * NFC_LANES;
   const uint32_t firstWindowSlot = 2 * finalLaneSlot;

   int rc;
   if ((rc = grow(ctx, ctx->wJobs, sizeof(NfcScanJob) * nJobs)) || (rc = grow(ctx, ctx->wChunks, sizeof(NfcScanChunk) * nChunks)) ||
       (rc = grow(ctx, ctx->wPoints, sizeof(NfcScanPoint) * (size_t)points)) || (rc = grow(ctx, ctx->wSeams, sizeof(NfcScanSeam) * nChunks)) ||
       (rc = grow(ctx, ctx->wChunkEdge, 4 * (size_t)nChunks)) || (rc = grow(ctx, ctx->wTiles, 4 * (size_t)tiles)) ||
       (rc = grow(ctx, ctx->wTileStats, sizeof(NfcScanTile) * (size_t)tiles)) ||
       (rc = grow(ctx, ctx->wCounters, 256)) || (rc = grow(ctx, ctx->wRepairs, sizeof(NfcScanChunk) * nChunks)) ||
       (rc = grow(ctx, ctx->wRepairsEnv, sizeof(NfcScanChunk) * nChunks)))
      return withoutTheMemory(rc);

   /* lanes: a first guess (one window per 8192 samples); the window kernel reports what it needs */
   uint32_t room = (uint32_t)(totalSamples / 8192) + 2 * nJobs + 64;

   /* records per lane slot (carry lanes, final lanes, one per window); ring and frame-assembly storage per carry lane,
    * final lane and per lane of the persistent waves that run the windows */
   const size_t storageLanes = (size_t)firstWindowSlot; /* (a speculative window's rings live in LDS; one that runs to the end leaves a copy in the save area) */

   auto growLanes = [&](uint32_t lanesWanted) -> int {
      const size_t lanes = ((size_t)lanesWanted + NFC_LANES - 1) / NFC_LANES * NFC_LANES;
      int r;
      if ((r = grow(ctx, ctx->wWindows, sizeof(NfcWindow) * lanes)) || (r = grow(ctx, ctx->wWorks, sizeof(NfcWork) * lanes)) ||
          (r = grow(ctx, ctx->vStates, sizeof(NfcStreamState) * lanes)) || (r = grow(ctx, ctx->vCold, sizeof(NfcStreamCold) * lanes)) ||
          (r = grow(ctx, ctx->wRunList, 4 * lanes)) ||
          (r = grow(ctx, ctx->vRings, sizeof(float) * (size_t)kRingBlockFloats * (storageLanes / NFC_LANES))) ||
          (r = grow(ctx, ctx->vBytes, (size_t)NFC_STREAM_BYTES * storageLanes)))
         return r;
      return NFCGPU_OK;
   };

   /* is there room already from an earlier, larger submission? */
   {
      /* (every lane buffer has to hold them: one that could not be grown last time - NFCGPU_ENOMEM, the submission then taken in
       * quarters - must not be asked for the room its neighbours got) */
      size_t have = ctx->wWindows.bytes / sizeof(NfcWindow);
      have = std::min(have, ctx->wWorks.bytes / sizeof(NfcWork));
      have = std::min(have, ctx->vStates.bytes / sizeof(NfcStreamState));
      have = std::min(have, ctx->vCold.bytes / sizeof(NfcStreamCold));
      have = std::min(have, ctx->wRunList.bytes / 4);
      if (have > (size_t)firstWindowSlot + room)
         room = (uint32_t)(have - firstWindowSlot - NFC_LANES);
   }

   if ((rc = growLanes(firstWindowSlot + room)))
      return withoutTheMemory(rc);

   /* staging sink for the lanes' chained frame records (lanes that turn out not to be live write theirs too): room
    * for four times the frame sink, at least 64 MiB; what does not fit is reported as dropped like any overflow */
   {
      size_t staging = (size_t)ctx->ownSinkWords * 16;
      if (staging < (64u << 20))
         staging = 64u << 20;
      if (staging > 0xFFFFFFF0ull * 4ull)
         staging = 0xFFFFFFF0ull * 4ull;
      if ((rc = grow(ctx, ctx->vSink, staging)) || (rc = grow(ctx, ctx->vSinkCtl, 16)))
         return withoutTheMemory(rc);
   }

   uint32_t *counters = (uint32_t *)ctx->wCounters.ptr;

   HIP_TRY(ctx, hipMemcpyAsync(ctx->wJobs.ptr, jobs.data(), sizeof(NfcScanJob) * nJobs, hipMemcpyHostToDevice, ctx->stream));
   HIP_TRY(ctx, hipMemcpyAsync(ctx->wChunks.ptr, chunks.data(), sizeof(NfcScanChunk) * nChunks, hipMemcpyHostToDevice, ctx->stream));
   HIP_TRY(ctx, hipMemsetAsync(counters, 0, 256, ctx->stream));
   HIP_TRY(ctx, hipMemsetAsync(ctx->vSinkCtl.ptr, 0, 16, ctx->stream));

   NfcScanArgs A;
   std::memset(&A, 0, sizeof(A));
   A.jobs = (NfcScanJob *)ctx->wJobs.ptr;
   A.nJobs = nJobs;
   A.chunks = (const NfcScanChunk *)ctx->wChunks.ptr;
   A.nChunks = nChunks;
   A.stride = stride;
   A.params = sp;
   A.states = ctx->dStates;
   A.points = (NfcScanPoint *)ctx->wPoints.ptr;
   A.seams = (NfcScanSeam *)ctx->wSeams.ptr;
   A.chunkEdge = (uint32_t *)ctx->wChunkEdge.ptr;
   A.tiles = (uint32_t *)ctx->wTiles.ptr;
   A.tileStats = (NfcScanTile *)ctx->wTileStats.ptr;
   A.windows = (NfcWindow *)ctx->wWindows.ptr;
   A.works = (NfcWork *)ctx->wWorks.ptr;
   A.finalLaneSlot = finalLaneSlot;
   A.firstWindowSlot = firstWindowSlot;
   A.windowRoom = room;
   A.windowCount = counters;
   A.rerunCount = counters + 1;
   A.runCount = counters + 2;
   A.runNext = counters + 3;
   A.runList = (uint32_t *)ctx->wRunList.ptr;
   A.repairs = (NfcScanChunk *)ctx->wRepairs.ptr;
   A.repairCount = counters + 7;
   A.repairsEnv = ctx->envelopeMax ? (NfcScanChunk *)ctx->wRepairsEnv.ptr : nullptr; /* (NFCGPU_ENVELOPE_KERNEL=0: one list, one kernel) */
   A.repairEnvCount = counters + 9;

   /* save area for lanes that run to the end of the submission (nfc_scan_launch.h): a few per stream */
   {
      const uint32_t saveRoom = 2 * nJobs + 1024;
      if ((rc = grow(ctx, ctx->vSaveRings, sizeof(float) * (size_t)(kRingBlockFloats / NFC_LANES) * saveRoom)) ||
          (rc = grow(ctx, ctx->vSaveBytes, (size_t)NFC_STREAM_BYTES * saveRoom)))
         return withoutTheMemory(rc);

      A.saveRings = (float *)ctx->vSaveRings.ptr;
      A.saveBytes = (uint8_t *)ctx->vSaveBytes.ptr;
      A.saveNext = counters + 8;
      A.saveRoom = saveRoom;
   }

   const NfcConfig *dCfg = ctx->dConfigs + config;

   /* NFCGPU_WINDOW_DEBUG: where the time of a submission goes (synchronises at every mark) */
   const bool debugStages = std::getenv("NFCGPU_WINDOW_DEBUG") != nullptr;
   auto stageBegan = entered;
   auto mark = [&](const char *what) {
      if (!debugStages)
         return;
      (void)hipStreamSynchronize(ctx->stream);
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[nfcgpu] windowed stage %-10s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - stageBegan).count());
      stageBegan = now;
   };

   mark("tables");

   /* scan */
   ProfiledLaunch pl {nullptr, nullptr};
   record_span(ctx, ctx->timedScan, pl, true);
   hipLaunchKernelGGL(nfc_scan_kernel, dim3((nChunks + NFC_LANES - 1) / NFC_LANES), dim3(NFC_LANES), 0, ctx->stream, dCfg, A);
   HIP_TRY(ctx, hipGetLastError());
   record_span(ctx, ctx->timedScan, pl, false);
   ctx->stats.scan_samples += totalSamples;

   /* windows (again with more room when the guess was short) */
   uint32_t nWindows = 0;

   mark("scan");

   /* (a grid has at most 65535 blocks in y: beyond 2^24 tiles in one job the kernel strides) */
   const uint32_t tilesGridY = (tilesMost + 255) / 256 > 65535u ? 65535u : (tilesMost + 255) / 256;

   /* a first run of the tile tests: how busy is each stream? Only a small submission is routed by that (below: `small`); a large
    * one gets its tile flags once, when the envelopes they are formed from are the true ones (3.8 ms for the 67 M tiles of config 5) */
   if (nJobs < NFC_LANES && !ctx->inBlocks)
   {
      hipLaunchKernelGGL(nfc_tiles_kernel, dim3(nJobs, tilesGridY), dim3(256), 0, ctx->stream, dCfg, A, tilesMost);
      HIP_TRY(ctx, hipGetLastError());
   }

   /* The front-end planes (below) are a walk of every chunk from its verified start state: 69 GB of stores for config 5, 32 ms
    * of a device that the rounds of second walks after the first leave nearly idle (a few thousand chunks each, as long as
    * their longest chain). Round 5: for a large submission that walk is started on a stream of its own (lowest priority) as soon as the first round's
    * second walks are queued - by then nine chunks in ten start from their true state -, the seam check and the envelope
    * walks note every start state they rewrite from then on (NfcScanArgs::planesStale), and those chunks' planes are written
    * again when the rounds are over. */
   const bool planesBeside = ctx->planesBeside && ctx->low != nullptr && totalSamples > (4u << 20);
   bool planesStarted = false;

   struct PlanesGuard
   {
      nfcgpu_ctx *ctx;
      bool running;
      ~PlanesGuard()
      {
         if (running)
            (void)hipStreamSynchronize(ctx->low); /* (whatever way the function is left: nobody reuses what the walk reads or writes while it runs) */
      }
   } planesGuard {ctx, false};

   if (planesBeside)
   {
      if ((rc = grow(ctx, ctx->wPlanes, (size_t)tiles * NFC_SCAN_TILE * 16u)) || (rc = grow(ctx, ctx->wPlaneChunks, sizeof(NfcScanChunk) * nChunks)) ||
          (rc = grow(ctx, ctx->wPlanesStale, 4u * (size_t)nChunks)))
         return withoutTheMemory(rc);

      HIP_TRY(ctx, hipMemsetAsync(ctx->wPlanesStale.ptr, 0, 4u * (size_t)nChunks, ctx->stream));
   }

   /* seams: chunks that did not start from the true state are walked again, a round at a time */

   for (uint32_t round = 0;; round++)
   {
      if (debugStages && std::atoi(std::getenv("NFCGPU_WINDOW_DEBUG")) >= 4)
      {
         /* which fields keep seams from verifying (host-side look at the records the seam check is about to judge) */
         std::vector<NfcScanSeam> sm(nChunks);
         HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
         HIP_TRY(ctx, hipMemcpy(sm.data(), ctx->wSeams.ptr, sizeof(NfcScanSeam) * nChunks, hipMemcpyDeviceToHost));
         uint32_t n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
         for (uint32_t j = 0; j < nJobs; j++)
            for (uint32_t k = 1; k < jobs[j].chunks; k++)
            {
               const NfcScanPoint &a = sm[jobs[j].firstChunk + k].start, &b = sm[jobs[j].firstChunk + k - 1].end;
               const bool env = std::memcmp(&a.env, &b.env, 4) != 0 || a.pulseFilter != b.pulseFilter;
               const bool n1 = std::memcmp(&a.n1, &b.n1, 4) != 0, mdev = std::memcmp(&a.mdev, &b.mdev, 4) != 0, avg = std::memcmp(&a.avg, &b.avg, 4) != 0;
               const bool peak = std::memcmp(&a.edgePeak, &b.edgePeak, 4) != 0, zone = ((a.zone ^ b.zone) & 0xFFu) != 0;
               const bool time = (a.zone & 0x100u) && (b.zone & 0x100u) && a.edgeTime != b.edgeTime;
               n[0] += env; n[1] += n1; n[2] += mdev; n[3] += avg; n[4] += peak; n[5] += zone; n[6] += time;
               n[7] += (n1 || mdev || avg || peak || zone) ? 1u : 0u;
            }
         std::fprintf(stderr, "[nfcgpu]    before round %u, seams that differ in: envelope / counter %u, n1 %u, deviation %u, average %u, edge peak %u, zone %u, known edge times %u; in anything but the envelope %u\n",
                      round, n[0], n[1], n[2], n[3], n[4], n[5], n[6], n[7]);
      }

      HIP_TRY(ctx, hipMemsetAsync(counters + 7, 0, 4, ctx->stream));
      HIP_TRY(ctx, hipMemsetAsync(counters + 9, 0, 4, ctx->stream));
      hipLaunchKernelGGL(nfc_seams_kernel, dim3((nJobs + 63) / 64), dim3(64), 0, ctx->stream, A, round == 0 ? 1u : 0u);
      HIP_TRY(ctx, hipGetLastError());

      /* chunks to walk again: every recurrence of them (A.repairs), the envelope tracker alone (A.repairsEnv: listed apart by
       * the seam check when the envelope kernel is on) */
      uint32_t word[3] = {0, 0, 0};
      HIP_TRY(ctx, hipMemcpyAsync(word, counters + 7, 12, hipMemcpyDeviceToHost, ctx->stream));

      /* (a small submission: how busy are its streams? the tile tests have counted) */
      const bool small = round == 0 && nJobs < NFC_LANES && !ctx->inBlocks;
      if (small)
         HIP_TRY(ctx, hipMemcpyAsync(jobs.data(), ctx->wJobs.ptr, sizeof(NfcScanJob) * nJobs, hipMemcpyDeviceToHost, ctx->stream));

      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      const uint32_t whole = word[0], alone = word[2];
      const uint32_t repairs = whole + alone;

      /* A few long busy streams: the passes the chain needs grow with the length of the submission (a frame that changes
       * the protocol timing is learnt one generation per pass), so it is decoded in blocks, each settled before the next.
       * Nothing has been touched yet (the scan only reads). */
      if (small)
      {
         uint32_t longest = 0;
         bool busy = false;

         for (uint32_t j = 0; j < nJobs; j++)
         {
            const uint64_t nTiles = ((uint64_t)jobs[j].count + NFC_SCAN_TILE - 1) / NFC_SCAN_TILE;
            busy = busy || (uint64_t)jobs[j].busyTiles * 100u > nTiles * ctx->busyPercent;
            longest = jobs[j].count > longest ? jobs[j].count : longest;
         }

         if (busy && longest > ctx->blockSamples)
            return run_in_blocks(ctx, config, items, stride);
      }

      if (debugStages)
         std::fprintf(stderr, "[nfcgpu]    seams round %u: %u chunks to walk again\n", round, repairs);

      if (!repairs)
         break;

      /* The chunks whose envelope tracker alone started wrong - from the second round on that is all of them: chains of chunks
       * that inherit a wrong envelope from each other, a chunk per round - go to a kernel that does nothing else, a wavefront
       * per chunk (nfc_envelope.hpp): a round then costs the tracker's own latency over one chunk instead of the scan kernel's
       * row machinery over it (9 ms of 32768 samples, however short the list). A list too long for a wave a chunk to pay (the
       * first round of a large submission: a quarter of its chunks, the scan kernel's 64 chunks per wave are the better use
       * of the machine) stays with the scan kernel's envelope-only branch (NFCGPU_ENVELOPE_KERNEL: the longest list the
       * envelope kernel is given). */
      if (debugStages && alone && alone <= ctx->envelopeMax)
         std::fprintf(stderr, "[nfcgpu]    ... %u of them the envelope tracker's alone, by the envelope kernel\n", alone);

      ProfiledLaunch pr {nullptr, nullptr};
      record_span(ctx, ctx->timedScan, pr, true);

      const bool byWaves = alone && alone <= ctx->envelopeMax;

      if (whole || (alone && !byWaves))
      {
         /* the scan kernel: the chunks walked whole, and a list of envelope-only ones too long for a wave each, in one launch */
         NfcScanArgs R = A;
         R.chunks = A.repairs;
         R.nChunks = whole;
         R.chunksMore = byWaves ? nullptr : A.repairsEnv;
         R.nChunksMore = byWaves ? 0u : alone;

         const uint32_t listedNow = R.nChunks + R.nChunksMore;

         hipLaunchKernelGGL(nfc_scan_kernel, dim3((listedNow + NFC_LANES - 1) / NFC_LANES), dim3(NFC_LANES), 0, ctx->stream, dCfg, R);
         HIP_TRY(ctx, hipGetLastError());
      }

      if (byWaves)
      {
         NfcScanArgs R = A;
         R.chunks = A.repairsEnv;
         R.nChunks = alone;
         R.followChains = alone <= ctx->envelopeFollowMax ? 1u : 0u;

         hipLaunchKernelGGL(nfc_envelope_kernel, dim3(alone), dim3(NFC_LANES), 0, ctx->stream, dCfg, R);
         HIP_TRY(ctx, hipGetLastError());
      }

      record_span(ctx, ctx->timedScan, pr, false);
      ctx->stats.scan_repairs += repairs;

      if (planesBeside && round == 0)
      {
         /* the planes of every chunk, beside the rounds to come */
         HIP_TRY(ctx, hipEventRecord(ctx->forkEvent, ctx->stream));
         HIP_TRY(ctx, hipStreamWaitEvent(ctx->low, ctx->forkEvent, 0));
         planesGuard.running = true;

         NfcScanArgs P = A;
         P.planes = (float *)ctx->wPlanes.ptr;
         P.chunks = (const NfcScanChunk *)ctx->wChunks.ptr; /* (the submission's chunk table as it is: the walk takes no notice of the repair marks) */
         P.nChunks = nChunks;
         /* a lane per piece, and the pieces of a chunk have to tile it: the largest multiple of the distance of the stored
          * points (every lane starts from one) that is no longer than NFCGPU_PLANES_BESIDE_PIECE and divides the chunk. (Until
          * round 6 the quotient was truncated: with a chunk that is no multiple of the piece - the default sizing gives any
          * multiple of 512 for totals between 2^28 and 2^30 samples - the tail of every chunk that was not walked again got no
          * planes at all.) A chunk is a multiple of the points' distance, so that distance always does. */
         P.planesPiece = ctx->planesBesidePiece / NFC_SCAN_POINT * NFC_SCAN_POINT;
         if (P.planesPiece > sp.chunkSamples)
            P.planesPiece = sp.chunkSamples / NFC_SCAN_POINT * NFC_SCAN_POINT;
         while (P.planesPiece > NFC_SCAN_POINT && sp.chunkSamples % P.planesPiece != 0u)
            P.planesPiece -= NFC_SCAN_POINT;
         if (P.planesPiece && sp.chunkSamples % P.planesPiece != 0u)
            P.planesPiece = 0u; /* (a chunk that is no multiple of the points' distance: a lane per chunk, from its start) */
         P.planesPerChunk = P.planesPiece ? sp.chunkSamples / P.planesPiece : 0u;

         const uint64_t lanesOfIt = (uint64_t)nChunks * (P.planesPerChunk ? P.planesPerChunk : 1u);

         ProfiledLaunch pp {nullptr, nullptr};
         record_span(ctx, ctx->timedPlanes, pp, true, ctx->low);
         hipLaunchKernelGGL(nfc_scan_planes_kernel, dim3((uint32_t)((lanesOfIt + NFC_LANES - 1) / NFC_LANES)), dim3(NFC_LANES), 0, ctx->low, dCfg, P);
         HIP_TRY(ctx, hipGetLastError());
         record_span(ctx, ctx->timedPlanes, pp, false, ctx->low);
         HIP_TRY(ctx, hipEventRecord(ctx->joinEvent, ctx->low));

         planesStarted = true;
         A.planesStale = (uint32_t *)ctx->wPlanesStale.ptr; /* from the next round on */
      }
   }
   hipLaunchKernelGGL(nfc_tiles_kernel, dim3(nJobs, tilesGridY), dim3(256), 0, ctx->stream, dCfg, A, tilesMost);
   HIP_TRY(ctx, hipGetLastError());

   mark("seams");

   /* The wave decoder takes the front end's results per sample instead of walking it again: a second walk of every
    * chunk from its verified start state (the repair form of the scan: no warm-up) writes them. */
   if (planesStarted)
   {
      /* the walk over all chunks has run beside the rounds: the chunks whose start state changed under it, again */
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->joinEvent, 0));
      planesGuard.running = false; /* (the main stream now waits for it) */

      HIP_TRY(ctx, hipMemsetAsync(counters + 11, 0, 4, ctx->stream));
      hipLaunchKernelGGL(nfc_planes_stale_kernel, dim3((nChunks + 255) / 256), dim3(256), 0, ctx->stream, A, (const NfcScanChunk *)ctx->wChunks.ptr, nChunks,
                         (NfcScanChunk *)ctx->wPlaneChunks.ptr, counters + 11);
      HIP_TRY(ctx, hipGetLastError());

      uint32_t again = 0;
      HIP_TRY(ctx, hipMemcpyAsync(&again, counters + 11, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

      A.planes = (float *)ctx->wPlanes.ptr;
      A.planesStale = nullptr;

      if (debugStages)
         std::fprintf(stderr, "[nfcgpu]    planes written beside the rounds; %u chunks of %u again\n", again, nChunks);

      if (again)
      {
         NfcScanArgs P = A;
         /* (a stored point per lane - 512 samples, from the points the rounds have left true -: a few thousand chunks a lane each
          * would take as long as one chunk's walk, 12 ms, with the device all but idle) */
         P.chunks = (const NfcScanChunk *)ctx->wPlaneChunks.ptr;
         P.nChunks = again;
         P.planesPiece = NFC_SCAN_POINT;
         P.planesPerChunk = sp.chunkSamples / NFC_SCAN_POINT;

         ProfiledLaunch pp {nullptr, nullptr};
         record_span(ctx, ctx->timedPlanes, pp, true);
         hipLaunchKernelGGL(nfc_scan_planes_kernel, dim3((uint32_t)(((uint64_t)again * P.planesPerChunk + NFC_LANES - 1) / NFC_LANES)), dim3(NFC_LANES), 0, ctx->stream, dCfg, P);
         HIP_TRY(ctx, hipGetLastError());
         record_span(ctx, ctx->timedPlanes, pp, false);
      }

      mark("planes");
   }
   else
   {
      /* A small submission - one a caller waits for - is walked a lane per stored point instead of a lane per chunk: 512 samples
       * instead of 4096 on the way of everything that follows (NFCGPU_PLANES_PIECE; a short capture: 1.1 -> 0.2 ms) */
      const uint32_t piece = totalSamples <= (4u << 20) ? ctx->planesPiece / NFC_SCAN_POINT * NFC_SCAN_POINT : 0u;

      std::vector<NfcScanChunk> all;

      if (piece)
      {
         for (uint32_t j = 0; j < nJobs; j++)
            for (uint32_t i = 0; i * piece < jobs[j].count; i++)
               all.push_back(NfcScanChunk {j, i | NFC_CHUNK_REPAIR});
      }
      else
      {
         all = chunks;
         for (NfcScanChunk &c: all)
            c.index |= NFC_CHUNK_REPAIR;
      }

      const uint32_t nPlaneLanes = (uint32_t)all.size();

      if ((rc = grow(ctx, ctx->wPlanes, (size_t)tiles * NFC_SCAN_TILE * 16u)) || (rc = grow(ctx, ctx->wPlaneChunks, sizeof(NfcScanChunk) * all.size())))
      {
         /* (the planes are 16 bytes per sample of the submission: 64 GiB for 4096 streams x 2^20) */
         return withoutTheMemory(rc);
      }

      HIP_TRY(ctx, hipMemcpyAsync(ctx->wPlaneChunks.ptr, all.data(), sizeof(NfcScanChunk) * all.size(), hipMemcpyHostToDevice, ctx->stream));

      A.planes = (float *)ctx->wPlanes.ptr;

      NfcScanArgs P = A;
      P.chunks = (const NfcScanChunk *)ctx->wPlaneChunks.ptr;
      P.nChunks = nPlaneLanes;
      P.planesPiece = piece;

      ProfiledLaunch pp {nullptr, nullptr};
      record_span(ctx, ctx->timedPlanes, pp, true);
      hipLaunchKernelGGL(nfc_scan_planes_kernel, dim3((nPlaneLanes + NFC_LANES - 1) / NFC_LANES), dim3(NFC_LANES), 0, ctx->stream, dCfg, P);
      HIP_TRY(ctx, hipGetLastError());
      record_span(ctx, ctx->timedPlanes, pp, false);
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); /* (the chunk list is a local) */

      mark("planes");
   }

   for (int attempt = 0; attempt < 2; attempt++)
   {
      hipLaunchKernelGGL(nfc_windows_kernel, dim3(nJobs), dim3(64), 0, ctx->stream, A);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipMemcpyAsync(&nWindows, counters, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

      if (nWindows <= room)
         break;

      room = nWindows + NFC_LANES;
      if ((rc = growLanes(firstWindowSlot + room)))
         return withoutTheMemory(rc);

      A.windows = (NfcWindow *)ctx->wWindows.ptr;
      A.works = (NfcWork *)ctx->wWorks.ptr;
         A.runList = (uint32_t *)ctx->wRunList.ptr;
      A.windowRoom = room;
      HIP_TRY(ctx, hipMemsetAsync(counters, 0, 4, ctx->stream));
   }

   mark("windows");

   NfcLaunch real = base_launch(ctx);

   NfcLaunch lanes;
   std::memset(&lanes, 0, sizeof(lanes));
   lanes.states = (NfcStreamState *)ctx->vStates.ptr;
   lanes.cold = (NfcStreamCold *)ctx->vCold.ptr;
   lanes.rings = (float *)ctx->vRings.ptr;
   lanes.bytes = (uint8_t *)ctx->vBytes.ptr;
   lanes.sink = (uint32_t *)ctx->vSink.ptr;
   lanes.sinkCtl = (uint32_t *)ctx->vSinkCtl.ptr;
   lanes.sinkWords = (uint32_t)(ctx->vSink.bytes / 4 > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : ctx->vSink.bytes / 4);
   if (ctx->stagingWords && lanes.sinkWords > ctx->stagingWords)
      lanes.sinkWords = ctx->stagingWords; /* (NFCGPU_STAGING_WORDS: the tests make it run full) */
   lanes.ringBlockFloats = kRingBlockFloats;
   lanes.works = (const NfcWork *)ctx->wWorks.ptr;
   lanes.windows = (NfcWindow *)ctx->wWindows.ptr;
   lanes.jobs = (const NfcScanJob *)ctx->wJobs.ptr;
   lanes.laneStats = counters + 4;
   lanes.uniformStride = stride;

   /* lanes */
   hipLaunchKernelGGL(nfc_carry_lanes_kernel, dim3(nJobs), dim3(NFC_LANES), 0, ctx->stream, A, real, lanes, 0u);
   HIP_TRY(ctx, hipGetLastError());

   const uint32_t windowBlocks = (nWindows + NFC_LANES - 1) / NFC_LANES;

   /* one lane per slot: the carry lanes (and, at the end, the lanes that regenerate a job's final state) */
   auto decodeSlots = [&](bool carry, uint32_t firstSlot, uint32_t slotCount, hipStream_t on) -> int {
      if (slotCount == 0)
         return NFCGPU_OK;

      NfcLaunch L = lanes;
      L.firstSlot = firstSlot;
      L.slotCount = slotCount;
      L.firstBlock = firstSlot / NFC_LANES;
      L.warmFront = carry ? 0u : NFC_WINDOW_WARM_FRONT;
      L.warmCorr = carry ? 0u : NFC_WINDOW_WARM_CORR;

      ProfiledLaunch wl {nullptr, nullptr};
      record_span(ctx, ctx->timedWave, wl, true, on);
      hipLaunchKernelGGL(nfc_wave_kernel, dim3(slotCount), dim3(NFC_LANES), 0, on, dCfg, L, A, carry ? 0u : 2u); /* a wave per lane */
      record_span(ctx, ctx->timedWave, wl, false, on);
      HIP_TRY(ctx, hipGetLastError());
      ctx->stats.launches++;
      return NFCGPU_OK;
   };

   /* the speculative windows on the run list: persistent waves */
   auto decodeWindows = [&](uint32_t runLanes) -> int {
      NfcLaunch L = lanes;
      L.warmFront = NFC_WINDOW_WARM_FRONT;
      L.warmCorr = NFC_WINDOW_WARM_CORR;

      ProfiledLaunch wl {nullptr, nullptr};
      record_span(ctx, ctx->timedWave, wl, true);
      hipLaunchKernelGGL(nfc_wave_kernel, dim3(runLanes), dim3(NFC_LANES), 0, ctx->stream, dCfg, L, A, 1u); /* a wave per run-list entry */
      record_span(ctx, ctx->timedWave, wl, false);
      HIP_TRY(ctx, hipGetLastError());
      ctx->stats.launches++;
      return NFCGPU_OK;
   };

   ProfiledLaunch pw {nullptr, nullptr};
   record_span(ctx, ctx->timedWindow, pw, true);

   uint32_t pass = 0;
   const bool debugPasses = std::getenv("NFCGPU_WINDOW_DEBUG") != nullptr;
   std::vector<hipEvent_t> passEvents; /* fork / join events in flight; back to the pool once the pass has been waited for */

   /* Whatever way this function is left while the side stream may still be running kernels of the pass (a launch that
    * failed after the fork, an event that could not be recorded): the side stream is waited for before anyone reuses the
    * staging slot or the lane buffers it reads, and the events of the pass go back to the pool. */
   struct SideGuard
   {
      nfcgpu_ctx *ctx;
      std::vector<hipEvent_t> *events;
      bool running;
      ~SideGuard()
      {
         if (running)
            (void)hipStreamSynchronize(ctx->side);
         for (hipEvent_t e: *events)
            ctx->eventPool.push_back(e);
         events->clear();
      }
   } sideGuard {ctx, &passEvents, false};

   for (;;)
   {
      const auto passBegan = std::chrono::steady_clock::now();

      if (nWindows)
      {
         HIP_TRY(ctx, hipMemsetAsync(counters + 2, 0, 8, ctx->stream)); /* run list: count and next */
         if (!ctx->longFirst)
         {
            hipLaunchKernelGGL(nfc_window_lanes_kernel, dim3(windowBlocks), dim3(NFC_LANES), 0, ctx->stream, dCfg, A, lanes, pass, 0u, 0u, 0xFFFFFFFFu);
            HIP_TRY(ctx, hipGetLastError());
         }
         else
         {
            /* longest lanes first: classes of 65536 samples and more, then halving down to NFCGPU_LONG_FIRST, then the rest */
            uint32_t hi = 0xFFFFFFFFu, lo = 65536u > ctx->longFirst ? 65536u : ctx->longFirst;

            for (uint32_t order = 1u;; order = 2u)
            {
               hipLaunchKernelGGL(nfc_window_lanes_kernel, dim3(windowBlocks), dim3(NFC_LANES), 0, ctx->stream, dCfg, A, lanes, pass, order, lo, hi);
               HIP_TRY(ctx, hipGetLastError());

               if (lo == 0u)
                  break;

               hi = lo;
               lo = lo / 2u >= ctx->longFirst ? lo / 2u : 0u;
            }
         }
      }

      /* how many lanes the list holds: the later passes of a submission list a few thousand, then a few dozen, of its windows -
       * the launch gets a grid of the list's length, not of the submission's window count */
      uint32_t runLanes = 0;
      if (nWindows)
      {
         HIP_TRY(ctx, hipMemcpyAsync(&runLanes, counters + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
         HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
         if (runLanes > nWindows)
            runLanes = nWindows;
      }

      /* the carry lanes (later passes: those the chain kernel sent on): from the stream's own state, which has not
       * been touched */
      if (pass > 0)
      {
         hipLaunchKernelGGL(nfc_carry_lanes_kernel, dim3(nJobs), dim3(NFC_LANES), 0, ctx->stream, A, real, lanes, pass);
         HIP_TRY(ctx, hipGetLastError());
      }

      /* carry lanes and speculative lanes side by side: no lane looks at what another one is doing while it runs
       * (nfc_lane_handover), and the longest lane of either kind can be most of the submission */
      if (ctx->sideMode == 0 || nWindows == 0 || runLanes == 0)
      {
         /* nothing to run beside (or NFCGPU_SIDE_STREAM=0): one stream */
         if ((rc = decodeSlots(true, 0, nJobs, ctx->stream)))
            return rc;
         if (nWindows && runLanes && (rc = decodeWindows(runLanes)))
            return rc;
      }
      else
      {
         /* events of this pass only (never recorded again while a wait on them may be pending) */
         hipEvent_t fork = ctx->sideMode == 2 ? take_event(ctx) : ctx->forkEvent;
         hipEvent_t join = ctx->sideMode == 2 ? take_event(ctx) : ctx->joinEvent;

         if (ctx->sideMode == 2)
         {
            passEvents.push_back(fork);
            passEvents.push_back(join);
         }

         HIP_TRY(ctx, hipEventRecord(fork, ctx->stream));
         HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, fork, 0));
         sideGuard.running = true;

         if ((rc = decodeSlots(true, 0, nJobs, ctx->side)))
            return rc;

         HIP_TRY(ctx, hipEventRecord(join, ctx->side));

         if (runLanes && (rc = decodeWindows(runLanes)))
            return rc;

         HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, join, 0));
         sideGuard.running = false; /* (the main stream now waits for it) */
      }

      HIP_TRY(ctx, hipMemsetAsync(counters + 1, 0, 4, ctx->stream));
      hipLaunchKernelGGL(nfc_chain_kernel, dim3((nJobs + 63) / 64), dim3(64), 0, ctx->stream, A, lanes, nJobs >= NFC_LANES ? ctx->maxPasses : ctx->maxPassesFew);
      HIP_TRY(ctx, hipGetLastError());

      uint32_t again = 0;
      HIP_TRY(ctx, hipMemcpyAsync(&again, counters + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

#if defined(NFCGPU_TUNING_KNOBS) || defined(NFCGPU_EMULATED_TEST_BUILD)
      /* (the tuning build: NFCGPU_DUMP_WINDOWS=<file> - the pieces of the first pass as they ended, five words each: job, start,
       * sample from which the piece was live, first sample not consumed, how it ended; profiles/tools/r06/item5_pieces_ab.py) */
      if (pass == 0 && nWindows)
         if (const char *dumpTo = std::getenv("NFCGPU_DUMP_WINDOWS"))
         {
            std::vector<NfcWindow> ws(nWindows);
            HIP_TRY(ctx, hipMemcpy(ws.data(), (const NfcWindow *)ctx->wWindows.ptr + firstWindowSlot, sizeof(NfcWindow) * ws.size(), hipMemcpyDeviceToHost));
            if (FILE *f = std::fopen(dumpTo, "ab"))
            {
               for (const NfcWindow &w: ws)
               {
                  const uint32_t rec[5] = {w.job, w.start, w.activate, w.stop, w.retired};
                  std::fwrite(rec, 4, 5, f);
               }
               std::fclose(f);
            }
         }
#endif

      if (debugPasses)
      {
         uint32_t ls[3] = {0, 0, 0}, tilesTaken = 0;
         HIP_TRY(ctx, hipMemcpy(ls, counters + 4, 12, hipMemcpyDeviceToHost));
         HIP_TRY(ctx, hipMemcpy(&tilesTaken, counters + 10, 4, hipMemcpyDeviceToHost));
         HIP_TRY(ctx, hipMemsetAsync(counters + 4, 0, 12, ctx->stream));
         HIP_TRY(ctx, hipMemsetAsync(counters + 10, 0, 4, ctx->stream));
         HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
         const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - passBegan).count();
         {
            /* (a library built with -DNFC_WAVE_PROFILE: shader cycles per phase of the wave decoder, nfc_wave.hpp) */
            uint32_t prof[16] = {0};
            HIP_TRY(ctx, hipMemcpy(prof, counters + 16, 64, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemsetAsync(counters + 16, 0, 64, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            uint64_t all = 0;
            for (uint32_t v: prof)
               all += v;
            if (all)
               std::fprintf(stderr, "[nfcgpu]    wave cycles x 2^10: boundary %u, tile load %u, values %u, search gates %u, commit %u, step %u, search step %u, set-up %u, "
                                    "prologue %u, locked gates %u, detectors shown their records %u (gates after it: %u), between %u, step: state in %u, machine %u, state out %u\n", prof[0], prof[1], prof[2], prof[3], prof[4], prof[5],
                            prof[6], prof[7], prof[8], prof[9], prof[10], prof[15], prof[11], prof[12], prof[13], prof[14]);
         }
         if (std::atoi(std::getenv("NFCGPU_WINDOW_DEBUG")) >= 2 && nWindows)
         {
            /* the lanes of the first stream as the pass left them */
            std::vector<NfcScanJob> jb(1);
            HIP_TRY(ctx, hipMemcpy(jb.data(), ctx->wJobs.ptr, sizeof(NfcScanJob), hipMemcpyDeviceToHost));
            std::vector<NfcWindow> ws(jb[0].windows);
            if (!ws.empty())
               HIP_TRY(ctx, hipMemcpy(ws.data(), (const NfcWindow *)ctx->wWindows.ptr + jb[0].firstWindow, sizeof(NfcWindow) * ws.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ws.size(); i++)
               std::fprintf(stderr, "[nfcgpu]    lane %3zu: start %8u live from %8u verify %8u stop %8u (%7u samples) retired %u to %3u, rerun %u live %u pub %u\n", i, ws[i].start,
                            ws[i].activate, ws[i].verify, ws[i].stop, ws[i].stop - ws[i].start, ws[i].retired, ws[i].handTo - jb[0].firstWindow, ws[i].rerun, ws[i].live,
                            ws[i].pubState);
         }
         if (std::atoi(std::getenv("NFCGPU_WINDOW_DEBUG")) >= 5 && nWindows)
         {
            /* why lanes are sent round again: what each lane marked for another run assumed (carry) against what it is told to assume (want) */
            std::vector<NfcScanJob> jb(nJobs);
            HIP_TRY(ctx, hipMemcpy(jb.data(), ctx->wJobs.ptr, sizeof(NfcScanJob) * nJobs, hipMemcpyDeviceToHost));
            uint32_t shown = 0;
            {
               /* tally over every lane marked: which fields of its assumption are to change */
               std::vector<NfcWindow> all(nWindows);
               HIP_TRY(ctx, hipMemcpy(all.data(), (const NfcWindow *)ctx->wWindows.ptr + firstWindowSlot, sizeof(NfcWindow) * all.size(), hipMemcpyDeviceToHost));
               uint64_t n[10] = {0}, samples = 0, over[4] = {0, 0, 0, 0};
               uint32_t longest = 0;
               for (const NfcWindow &w: all)
               {
                  if (!w.rerun)
                     continue;
                  const NfcCarry &a = w.carry, &b = w.want;
                  bool tim = false, wait = false;
                  for (int t = 0; t < 4; t++)
                  {
                     tim = tim || a.tim[t].lastCommand != b.tim[t].lastCommand || a.tim[t].maxFrameSize != b.tim[t].maxFrameSize || a.tim[t].protoGuardTime != b.tim[t].protoGuardTime;
                     wait = wait || a.tim[t].protoWaitingTime != b.tim[t].protoWaitingTime;
                  }
                  const bool chained = a.chainedA != b.chainedA, carrier = (a.carrierOn != 0) != (b.carrierOn != 0) || (a.carrierOff != 0) != (b.carrierOff != 0);
                  const bool emit = a.emitValid != b.emitValid || a.emitClock != b.emitClock;
                  const bool pulses = a.pulsesF[0] != b.pulsesF[0] || a.pulsesF[1] != b.pulsesF[1] || std::memcmp(a.thrF, b.thrF, 8) != 0;
                  const bool records = std::memcmp(&a.search, &b.search, sizeof(a.search)) != 0;
                  n[0]++; n[1] += chained; n[2] += carrier; n[3] += emit; n[4] += tim; n[5] += wait; n[6] += pulses; n[7] += records;
                  n[8] += !(chained || carrier || emit || tim || wait || pulses || records) ? 1 : 0;
                  n[9] += (chained && !(carrier || tim || wait || pulses || records)) ? 1 : 0;
                  samples += w.stop - w.start;
                  const uint32_t len = w.stop - w.start;
                  longest = len > longest ? len : longest;
                  over[0] += len >= 65536u; over[1] += len >= 131072u; over[2] += len >= 196608u; over[3] += len >= 262144u;
               }
               {
                  /* the longest of them, and the streams they belong to */
                  std::vector<const NfcWindow *> byLen;
                  for (const NfcWindow &w: all)
                     if (w.rerun)
                        byLen.push_back(&w);
                  std::sort(byLen.begin(), byLen.end(), [](const NfcWindow *a, const NfcWindow *b) { return a->stop - a->start > b->stop - b->start; });
                  for (size_t i = 0; i < byLen.size() && i < 6; i++)
                     std::fprintf(stderr, "[nfcgpu]    long lane: job %u start %u activate %u stop %u (%u samples) retired %u handTo %u noHand %u\n", byLen[i]->job, byLen[i]->start, byLen[i]->activate,
                                  byLen[i]->stop, byLen[i]->stop - byLen[i]->start, byLen[i]->retired, byLen[i]->handTo, byLen[i]->noHand);
               }
               std::fprintf(stderr, "[nfcgpu]    ... of them %llu ran 65536 samples and more, %llu 131072+, %llu 196608+, %llu 262144+; the longest %u\n",
                            (unsigned long long)over[0], (unsigned long long)over[1], (unsigned long long)over[2], (unsigned long long)over[3], longest);
               std::fprintf(stderr, "[nfcgpu]    lanes to run again %llu (%llu samples as they last ran): chainedA %llu (and nothing else but the carrier record: %llu), carrier on/off %llu, "
                                    "carrier record %llu, command / frame size / guard %llu, waiting time %llu, NFC-F pulse memory %llu, detector records %llu, same assumption (sent on past a hand-over) %llu\n",
                            (unsigned long long)n[0], (unsigned long long)samples, (unsigned long long)n[1], (unsigned long long)n[9], (unsigned long long)n[2], (unsigned long long)n[3],
                            (unsigned long long)n[4], (unsigned long long)n[5], (unsigned long long)n[6], (unsigned long long)n[7], (unsigned long long)n[8]);
            }
            for (uint32_t j = 0; j < nJobs && shown < 24 && pass >= 3; j++)
            {
               if (!(jb[j].status & NFC_JOB_RERUN) || !jb[j].windows)
                  continue;
               std::vector<NfcWindow> ws(jb[j].windows);
               HIP_TRY(ctx, hipMemcpy(ws.data(), (const NfcWindow *)ctx->wWindows.ptr + jb[j].firstWindow, sizeof(NfcWindow) * ws.size(), hipMemcpyDeviceToHost));
               for (size_t i = 0; i < ws.size(); i++)
               {
                  const NfcWindow &w = ws[i];
                  if (!w.rerun)
                     continue;
                  shown++;
                  std::fprintf(stderr, "[nfcgpu]    job %u lane %zu/%zu start %u stop %u retired %u noHand %u handTo %u:", j, i, ws.size(), w.start, w.stop, w.retired, w.noHand, w.handTo);
                  const NfcCarry &a = w.carry, &b = w.want;
                  if (a.chainedA != b.chainedA) std::fprintf(stderr, " chainedA %u->%u", a.chainedA, b.chainedA);
                  if ((a.carrierOn != 0) != (b.carrierOn != 0)) std::fprintf(stderr, " carrierOn %u->%u", a.carrierOn, b.carrierOn);
                  if ((a.carrierOff != 0) != (b.carrierOff != 0)) std::fprintf(stderr, " carrierOff %u->%u", a.carrierOff, b.carrierOff);
                  if (a.emitValid != b.emitValid || a.emitClock != b.emitClock) std::fprintf(stderr, " emit %u/%u->%u/%u (tracked %u)", a.emitValid, a.emitClock, b.emitValid, b.emitClock, w.tracked);
                  for (int t = 0; t < 4; t++)
                  {
                     if (a.tim[t].lastCommand != b.tim[t].lastCommand) std::fprintf(stderr, " cmd[%d] %u->%u", t, a.tim[t].lastCommand, b.tim[t].lastCommand);
                     if (a.tim[t].maxFrameSize != b.tim[t].maxFrameSize) std::fprintf(stderr, " maxFrame[%d] %u->%u", t, a.tim[t].maxFrameSize, b.tim[t].maxFrameSize);
                     if (a.tim[t].protoGuardTime != b.tim[t].protoGuardTime) std::fprintf(stderr, " guard[%d] %u->%u", t, a.tim[t].protoGuardTime, b.tim[t].protoGuardTime);
                     if (a.tim[t].protoWaitingTime != b.tim[t].protoWaitingTime) std::fprintf(stderr, " wait[%d] %u->%u", t, a.tim[t].protoWaitingTime, b.tim[t].protoWaitingTime);
                  }
                  for (int i2 = 0; i2 < 2; i2++)
                     if (a.pulsesF[i2] != b.pulsesF[i2] || std::memcmp(&a.thrF[i2], &b.thrF[i2], 4)) std::fprintf(stderr, " F%d pulses %u->%u thr %g->%g", i2, a.pulsesF[i2], b.pulsesF[i2], a.thrF[i2], b.thrF[i2]);
                  if (std::memcmp(&a.search, &b.search, sizeof(a.search))) std::fprintf(stderr, " records");
                  std::fprintf(stderr, "\n");
               }
            }
         }
         /* (lane-steps: samples handled one by one by the step machine; tiles: what the lanes took in all, warm-ups included) */
         std::fprintf(stderr, "[nfcgpu] windowed pass %u: %u lanes, %u tiles of %llu (%.1f tiles per us), %llu lane-steps (%.2f per sample), longest lane %u steps, %u streams unsettled, %.1f ms\n",
                      pass, ls[2], tilesTaken, (unsigned long long)((totalSamples + NFC_SCAN_TILE - 1) / NFC_SCAN_TILE), ms > 0.0 ? (double)tilesTaken / (ms * 1000.0) : 0.0,
                      (unsigned long long)ls[0] * NFC_SCAN_TILE, (double)ls[0] * NFC_SCAN_TILE / (double)totalSamples, ls[1] * NFC_SCAN_TILE, again, ms);
      }

      for (hipEvent_t e: passEvents)
         ctx->eventPool.push_back(e); /* (the stream has been synchronised above) */
      passEvents.clear();

      ctx->stats.window_passes++;
      pass++;

      if (!again || !nWindows)
         break;
   }

   mark("passes");

   /* a library whose wave decoder was built with -DNFC_WAVE_VERIFY (Makefile: libnfcgpu_verify.so) has decoded every tile twice
    * and counted (csrc/nfc_wave.hip); a product build leaves the words at zero */
   if (std::getenv("NFCGPU_WAVE_VERIFY_REPORT"))
   {
      uint32_t v[4] = {0, 0, 0, 0};
      HIP_TRY(ctx, hipMemcpyAsync(v, counters + 12, 16, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      std::fprintf(stderr, "[nfcgpu] wave verify: %u tiles decoded twice (bulk paths / step machine alone), %u differ", v[0], v[1]);
      if (v[1])
         std::fprintf(stderr, " (the first at stream position %u of lane slot %u)", v[2], v[3]);
      std::fprintf(stderr, "\n");
   }

   /* The lanes chain their frame records in a staging sink that is sized from an estimate and written by every lane of
    * every pass, live in the end or not. Should it have run full, frames of live lanes may be among the ones that did not
    * fit: nothing of the streams has been touched yet, so the submission is decoded by the sequential kernels instead. */
   {
      uint32_t stagingCtl[2] = {0, 0};
      HIP_TRY(ctx, hipMemcpyAsync(stagingCtl, ctx->vSinkCtl.ptr, 8, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

      if (stagingCtl[1])
      {
         HIP_TRY(ctx, hipMemsetAsync(ctx->vSinkCtl.ptr, 0, 16, ctx->stream));
         ctx->stats.fallback_streams += nJobs;
         return launch_sequential(ctx, config, items, stride);
      }
   }

   /* the state a stream is left in: its last lane's, run once more with storage of its own */
   hipLaunchKernelGGL(nfc_final_lanes_kernel, dim3((nJobs + 63) / 64), dim3(64), 0, ctx->stream, dCfg, A, lanes);
   HIP_TRY(ctx, hipGetLastError());

   if ((rc = decodeSlots(false, finalLaneSlot, nJobs, ctx->stream)))
      return rc;

   record_span(ctx, ctx->timedWindow, pw, false);

   hipLaunchKernelGGL(nfc_finish_kernel, dim3(nJobs), dim3(NFC_LANES), 0, ctx->stream, A, real, lanes);
   HIP_TRY(ctx, hipGetLastError());

   HIP_TRY(ctx, hipMemcpyAsync(jobs.data(), ctx->wJobs.ptr, sizeof(NfcScanJob) * nJobs, hipMemcpyDeviceToHost, ctx->stream));
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

   mark("finish");

   if (debugStages && std::atoi(std::getenv("NFCGPU_WINDOW_DEBUG")) >= 3 && nWindows)
   {
      /* how much of what the speculative lanes decoded ended up in the stream (the rest was overrun by a lane that could
       * not hand over, or decoded again in a later pass) */
      std::vector<NfcWindow> ws(nWindows);
      HIP_TRY(ctx, hipMemcpy(ws.data(), (const NfcWindow *)ctx->wWindows.ptr + firstWindowSlot, sizeof(NfcWindow) * ws.size(), hipMemcpyDeviceToHost));
      uint64_t all = 0, live = 0, liveLanes = 0;
      for (const NfcWindow &w: ws)
      {
         all += w.stop - w.start;
         if (w.live)
         {
            live += w.stop - w.start;
            liveLanes++;
         }
      }
      std::fprintf(stderr, "[nfcgpu] speculative lanes: %zu, %llu samples as they last ran; live in the end: %llu lanes, %llu samples (%.1f %%); submission: %llu samples\n", ws.size(),
                   (unsigned long long)all, (unsigned long long)liveLanes, (unsigned long long)live, all ? 100.0 * (double)live / (double)all : 0.0, (unsigned long long)totalSamples);
   }

   ctx->stats.windows += nWindows + nJobs;
   ctx->stats.samples += totalSamples;
   ctx->dirty = true;

   std::vector<WindowedItem> fallback;

   for (uint32_t j = 0; j < nJobs; j++)
   {
      if (jobs[j].status & NFC_JOB_INVALID)
         fallback.push_back(items[j]);
      else
      {
         ctx->streams[items[j].slot].clock += items[j].count;
         ctx->stats.windowed_streams++;
      }
   }

   ctx->stats.fallback_streams += fallback.size();

   mark("return");

   if (!fallback.empty())
   {
      ctx->stats.samples -= 0; /* launch_demod counts the samples of the streams it decodes */
      uint64_t again = 0;
      for (const WindowedItem &it: fallback)
         again += it.count;
      ctx->stats.samples -= again;
      return launch_sequential(ctx, config, fallback, stride);
   }

   return NFCGPU_OK;
}

/* the next staging slot, idle and at least `bytes` large */
int stage_acquire(nfcgpu_ctx *ctx, size_t bytes, nfcgpu_ctx::StageSlot **out)
{
   nfcgpu_ctx::StageSlot &slot = ctx->stage[ctx->stageNext];
   ctx->stageNext ^= 1u;

   if (slot.busy)
   {
      /* the launches of the submission before the last one: long done in a steady stream of submissions */
      HIP_TRY(ctx, hipEventSynchronize(slot.done));
      slot.busy = false;
   }

   if (!slot.done)
      HIP_TRY(ctx, hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));

   if (bytes > slot.bytes)
   {
      if (slot.d)
         (void)hipFree(slot.d);
      if (slot.h)
         (void)hipHostFree(slot.h);

      slot.d = nullptr;
      slot.h = nullptr;
      slot.bytes = 0;

      const size_t want = bytes + bytes / 2;
      if (hipMalloc((void **)&slot.d, want) != hipSuccess)
         return fail(ctx, NFCGPU_ENOMEM, "staging buffer allocation failed");
      if (hipHostMalloc((void **)&slot.h, want, hipHostMallocDefault) != hipSuccess)
      {
         (void)hipFree(slot.d);
         slot.d = nullptr;
         return fail(ctx, NFCGPU_ENOMEM, "pinned staging buffer allocation failed");
      }

      slot.bytes = want;
   }

   *out = &slot;
   return NFCGPU_OK;
}

/* everything that reads the slot has been enqueued */
void stage_release(nfcgpu_ctx *ctx, nfcgpu_ctx::StageSlot *slot)
{
   if (slot && hipEventRecord(slot->done, ctx->stream) == hipSuccess)
      slot->busy = true;
   else if (slot)
      (void)hipStreamSynchronize(ctx->stream);
}

void drain_sink(nfcgpu_ctx *ctx, uint64_t cursor)
{
   const uint64_t valid = cursor < ctx->sinkWords ? cursor : ctx->sinkWords;
   uint64_t pos = 0;

   while (pos < valid && pos + NFC_FRAME_MAX_WORDS <= ctx->sinkWords)
   {
      const uint32_t *w = ctx->hSink.data() + pos;
      const uint32_t len = w[8] > NFC_STREAM_BYTES ? NFC_STREAM_BYTES : w[8];
      const uint32_t id = w[0];

      if (id < ctx->streams.size())
      {
         StreamInfo &si = ctx->streams[id];

         nfcgpu_frame f;
         std::memset(&f, 0, sizeof(f));
         f.stream_id = id;
         f.tech_type = w[1];
         f.frame_type = w[2];
         f.frame_flags = w[3];
         f.frame_phase = w[4];
         f.frame_rate = w[5];
         f.sample_start = w[6];
         f.sample_end = w[7];
         f.sample_rate = si.params.sample_rate;
         f.length = len;
         std::memcpy(f.data, w + NFC_FRAME_HEADER_WORDS, len);

         if (si.open)
            si.queue.push_back(f);

         ctx->stats.frames++;
      }

      pos += NFC_FRAME_HEADER_WORDS + ((len + 3) >> 2);
   }
}

}


/* ------------------------------------------------------------------------------------------ */
/* RCCL, looked up at run time (a host with a single GPU does not need it)                       */
/* ------------------------------------------------------------------------------------------ */
/* ncclUniqueId is passed by value: 128 bytes */
struct IdByValue
{
   char internal[NFCGPU_UNIQUE_ID_BYTES];
};

#ifdef NFCGPU_EMULATED_TEST_BUILD
struct FakeNcclId
{
   char internal[NFCGPU_UNIQUE_ID_BYTES];
};
extern "C" {
int fake_ncclGetUniqueId(void *id);
int fake_ncclCommInitRank(void **comm, int nRanks, FakeNcclId id, int rank);
int fake_ncclCommDestroy(void *comm);
int fake_ncclAllGather(const void *send, void *recv, size_t count, int type, void *comm, void *stream);
int fake_ncclBroadcast(const void *send, void *recv, size_t count, int type, int root, void *comm, void *stream);
int fake_ncclGroupStart();
int fake_ncclGroupEnd();
}
#endif

namespace {

struct Rccl
{
   void *handle = nullptr;
   int (*getUniqueId)(void *) = nullptr;
   int (*commInitRank)(void **, int, IdByValue, int) = nullptr;
   int (*commDestroy)(void *) = nullptr;
   int (*allGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
   int (*broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
   int (*groupStart)() = nullptr;
   int (*groupEnd)() = nullptr;
   const char *(*getErrorString)(int) = nullptr;
};

Rccl *rccl()
{
   static Rccl r;
   static bool tried = false;

   if (!tried)
   {
      tried = true;
#ifdef NFCGPU_EMULATED_TEST_BUILD
      /* the emulated test build has no RCCL; with NFCGPU_FAKE_RCCL=1 it takes an in-process stand-in (tests/hostsim/
       * fake_rccl.cpp: ranks are threads of one process) so that the rank logic of the gather runs without GPUs */
      if (std::getenv("NFCGPU_FAKE_RCCL"))
      {
         r.handle = (void *)&r;
         r.getUniqueId = fake_ncclGetUniqueId;
         r.commInitRank = (int (*)(void **, int, IdByValue, int))fake_ncclCommInitRank;
         r.commDestroy = fake_ncclCommDestroy;
         r.allGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))fake_ncclAllGather;
         r.broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))fake_ncclBroadcast;
         r.groupStart = fake_ncclGroupStart;
         r.groupEnd = fake_ncclGroupEnd;
         r.getErrorString = nullptr;
      }
#else
      for (const char *name: {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      {
         r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
         if (r.handle)
            break;
      }

      if (r.handle)
      {
         r.getUniqueId = (int (*)(void *))dlsym(r.handle, "ncclGetUniqueId");
         r.commInitRank = (int (*)(void **, int, IdByValue, int))dlsym(r.handle, "ncclCommInitRank");
         r.commDestroy = (int (*)(void *))dlsym(r.handle, "ncclCommDestroy");
         r.allGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(r.handle, "ncclAllGather");
         r.broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(r.handle, "ncclBroadcast");
         r.groupStart = (int (*)())dlsym(r.handle, "ncclGroupStart");
         r.groupEnd = (int (*)())dlsym(r.handle, "ncclGroupEnd");
         r.getErrorString = (const char *(*)(int))dlsym(r.handle, "ncclGetErrorString");
      }
#endif
   }

   return (r.handle && r.getUniqueId && r.commInitRank && r.commDestroy && r.allGather && r.broadcast && r.groupStart && r.groupEnd) ? &r : nullptr;
}

}

namespace {

/* what a context owns besides the slots: streams, events, the work buffers of the time-parallel path */
void release_workspace(nfcgpu_ctx *ctx)
{
   if (ctx->stream)
      (void)hipStreamSynchronize(ctx->stream);
   if (ctx->side)
      (void)hipStreamSynchronize(ctx->side);
   if (ctx->low)
      (void)hipStreamSynchronize(ctx->low);

   for (auto *list: {&ctx->timed, &ctx->timedScan, &ctx->timedWindow, &ctx->timedWave, &ctx->timedPlanes})
   {
      for (auto &pl: *list)
      {
         (void)hipEventDestroy(pl.start);
         (void)hipEventDestroy(pl.stop);
      }
      list->clear();
   }
   for (auto e: ctx->eventPool)
      (void)hipEventDestroy(e);
   ctx->eventPool.clear();

   for (nfcgpu_ctx::DevBuf *b: {&ctx->wRepairs, &ctx->wRepairsEnv, &ctx->wJobs, &ctx->wChunks, &ctx->wPoints, &ctx->wSeams, &ctx->wChunkEdge, &ctx->wTiles, &ctx->wTileStats, &ctx->wWindows, &ctx->wRunList,
                                &ctx->wWorks, &ctx->wCounters, &ctx->vStates, &ctx->vCold, &ctx->vRings, &ctx->vBytes, &ctx->vSink, &ctx->vSinkCtl, &ctx->vSaveRings, &ctx->vSaveBytes,
                                &ctx->wPlanes, &ctx->wPlaneChunks})
   {
      if (b->ptr)
         (void)hipFree(b->ptr);
      b->ptr = nullptr;
      b->bytes = 0;
   }

   if (ctx->epoch)
      (void)hipEventDestroy(ctx->epoch);
   ctx->epoch = nullptr;
   if (ctx->forkEvent)
      (void)hipEventDestroy(ctx->forkEvent);
   if (ctx->joinEvent)
      (void)hipEventDestroy(ctx->joinEvent);
   if (ctx->side)
      (void)hipStreamDestroy(ctx->side);
   if (ctx->low)
      (void)hipStreamDestroy(ctx->low);
   ctx->low = nullptr;
   if (ctx->stream)
      (void)hipStreamDestroy(ctx->stream);
   ctx->forkEvent = ctx->joinEvent = nullptr;
   ctx->side = ctx->stream = nullptr;
}

/* A profiled service that never resets its statistics must not collect a span per launch for ever: once the list is long, the
 * spans that end before the most recent ones begin - their union can no longer change - are folded into a running total. */
void fold_wave_spans(nfcgpu_ctx *ctx)
{
   std::vector<std::pair<float, float>> &v = ctx->waveSpans;

   if (v.size() < 4096)
      return;

   std::sort(v.begin(), v.end());

   /* the list becomes the union of its spans; an interval of that union that ends before the most recent 64 spans begin can
    * no longer grow (launches of one submission overlap, those of different submissions do not) and goes into the total */
   const float horizon = v[v.size() - 64].first;
   std::vector<std::pair<float, float>> open;
   double busy = 0.0;
   float lo = v[0].first, hi = v[0].second;

   for (size_t i = 1; i < v.size(); i++)
   {
      if (v[i].first > hi)
      {
         if (hi < horizon)
            busy += hi - lo;
         else
            open.push_back(std::make_pair(lo, hi));
         lo = v[i].first;
         hi = v[i].second;
      }
      else if (v[i].second > hi)
         hi = v[i].second;
   }

   open.push_back(std::make_pair(lo, hi));

   ctx->waveBusyMs += busy;
   v.swap(open);
}

/* HIP-event spans recorded since the last call -> milliseconds in the context's statistics; the events go back to the pool.
 * The streams the events were recorded on must have been waited for. */
void collect_timings(nfcgpu_ctx *ctx)
{
   struct Into
   {
      std::vector<ProfiledLaunch> *list;
      double *ms;
      uint64_t *count;
   } all[] = {{&ctx->timed, &ctx->stats.kernel_ms, nullptr},
              {&ctx->timedScan, &ctx->stats.scan_ms, nullptr},
              {&ctx->timedWindow, &ctx->stats.window_ms, nullptr},
              {&ctx->timedWave, &ctx->stats.wave_ms, &ctx->stats.wave_launches},
              {&ctx->timedPlanes, &ctx->stats.planes_ms, nullptr}};

   for (Into &into: all)
   {
      for (auto &pl: *into.list)
      {
         float ms = 0;
         if (hipEventElapsedTime(&ms, pl.start, pl.stop) == hipSuccess)
         {
            *into.ms += ms;
            if (into.count)
               ++*into.count;

            /* the wave decoder's launches run on two streams side by side: where they lie in time, for the time the kernel
             * was running at all (nfcgpu_stats::wave_busy_ms) */
            float at = 0;
            if (into.list == &ctx->timedWave && ctx->epoch && hipEventElapsedTime(&at, ctx->epoch, pl.start) == hipSuccess)
               ctx->waveSpans.push_back(std::make_pair(at, at + ms));
         }
         ctx->eventPool.push_back(pl.start);
         ctx->eventPool.push_back(pl.stop);
      }
      into.list->clear();
   }

   fold_wave_spans(ctx);
}

int run_rows(nfcgpu_ctx *ctx, uint32_t first, uint32_t count, const uint8_t *devBase, uint64_t devPitch, uint32_t n, uint32_t stride);

}

extern "C" {

void nfcgpu_default_params(nfcgpu_params *p)
{
   if (!p)
      return;

   std::memset(p, 0, sizeof(*p));
   p->sample_rate = 0;
   p->tech_mask = NFCGPU_TECH_A | NFCGPU_TECH_B | NFCGPU_TECH_F | NFCGPU_TECH_V;
   p->power_level_threshold = 0.01f;

   const float corr[4] = {0.75f, 0.50f, 0.50f, 0.50f};
   const float lo[4] = {0.90f, 0.10f, 0.10f, 0.90f};
   const float hi[4] = {1.00f, 0.90f, 0.90f, 1.00f};

   for (int t = 0; t < 4; t++)
   {
      p->corr_threshold[t] = corr[t];
      p->min_modulation_depth[t] = lo[t];
      p->max_modulation_depth[t] = hi[t];
   }
}

int nfcgpu_init(int device, const nfcgpu_options *options, nfcgpu_ctx **out)
{
   if (!out)
      return NFCGPU_EINVAL;

   *out = nullptr;

   int count = 0;
   if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
      return NFCGPU_ENODEV;

   if (hipSetDevice(device) != hipSuccess)
      return NFCGPU_ENODEV;

   nfcgpu_ctx *ctx = new (std::nothrow) nfcgpu_ctx();
   if (!ctx)
      return NFCGPU_ENOMEM;

   ctx->device = device;

   const char *generic = std::getenv("NFCGPU_GENERIC_KERNELS");
   ctx->genericOnly = generic && generic[0] == '1';

   /* Tuning switches of the time-parallel path. They are experiment switches, not configuration (round 6: the product library
    * used to read all nineteen from the environment): libnfcgpu.so runs on the defaults of nfcgpu_ctx and does not look at them.
    * A build of this file with -DNFCGPU_TUNING_KNOBS (`make tuning`: libnfcgpu_tuning.so, the same kernels; and the emulated test
    * build) takes them from the environment - what the tests use to force every path on small inputs and what the sweeps under
    * profiles/ were made with. What the product reads: NFCGPU_WINDOW_DEBUG (stage log on stderr), NFCGPU_GENERIC_KERNELS (the
    * any-rate kernels at the compiled-in rate too) and, in the shim, NFCGPU_DEVICE / NFCGPU_MAX_STREAMS / NFCGPU_SHIM_BLOCK(_MS). */
#if defined(NFCGPU_TUNING_KNOBS) || defined(NFCGPU_EMULATED_TEST_BUILD)
   auto tuning = [](const char *name) -> const char * { return std::getenv(name); };
#else
   auto tuning = [](const char *) -> const char * { return nullptr; };
#endif
   auto knob = [&](const char *name, uint32_t fallback) -> uint32_t {
      const char *v = tuning(name);
      return v && v[0] ? (uint32_t)std::strtoul(v, nullptr, 10) : fallback;
   };

   ctx->windowed = knob("NFCGPU_WINDOWED", 1) != 0;
   ctx->windowedMinSamples = knob("NFCGPU_WINDOWED_MIN", ctx->windowedMinSamples);
   ctx->scanChunkFixed = tuning("NFCGPU_SCAN_CHUNK") != nullptr && tuning("NFCGPU_SCAN_CHUNK")[0] != 0;
   ctx->scanChunk = knob("NFCGPU_SCAN_CHUNK", ctx->scanChunk) / NFC_SCAN_POINT * NFC_SCAN_POINT;
   ctx->scanLanes = knob("NFCGPU_SCAN_LANES", ctx->scanLanes);
   if (ctx->scanLanes == 0u)
      ctx->scanLanes = 1u;
   ctx->scanWarm = knob("NFCGPU_SCAN_WARM", ctx->scanWarm) / NFC_SCAN_POINT * NFC_SCAN_POINT;
   ctx->maxPasses = knob("NFCGPU_WINDOW_PASSES", ctx->maxPasses);
   ctx->maxPassesFew = knob("NFCGPU_WINDOW_PASSES_FEW", knob("NFCGPU_WINDOW_PASSES", ctx->maxPassesFew));
   ctx->soloSamples = knob("NFCGPU_SOLO_SAMPLES", ctx->soloSamples);
   ctx->stagingWords = knob("NFCGPU_STAGING_WORDS", ctx->stagingWords);
   ctx->lanesWanted = knob("NFCGPU_LANES_WANTED", ctx->lanesWanted);
   ctx->longFirst = knob("NFCGPU_LONG_FIRST", ctx->longFirst);
   ctx->envelopeMax = knob("NFCGPU_ENVELOPE_KERNEL", ctx->envelopeMax);
   ctx->envelopeFollowMax = knob("NFCGPU_ENVELOPE_FOLLOW", ctx->envelopeFollowMax);
   ctx->planesPiece = knob("NFCGPU_PLANES_PIECE", ctx->planesPiece);
   ctx->planesBeside = knob("NFCGPU_PLANES_BESIDE", ctx->planesBeside);
   ctx->planesBesidePiece = knob("NFCGPU_PLANES_BESIDE_PIECE", ctx->planesBesidePiece);
   ctx->cutMax = knob("NFCGPU_CUT_MAX", ctx->cutMax);
   if (ctx->cutMax < NFC_WINDOW_CUT)
      ctx->cutMax = NFC_WINDOW_CUT;
   ctx->sideMode = knob("NFCGPU_SIDE_STREAM", ctx->sideMode);
   ctx->blockSamples = knob("NFCGPU_BLOCK_SAMPLES", ctx->blockSamples) / NFC_SCAN_POINT * NFC_SCAN_POINT;
   if (ctx->blockSamples < 65536u)
      ctx->blockSamples = 65536u;

   if (ctx->scanWarm < NFC_SCAN_POINT)
      ctx->scanWarm = NFC_SCAN_POINT;
   if (ctx->scanChunk < ctx->scanWarm + NFC_SCAN_POINT)
      ctx->scanChunk = ctx->scanWarm + NFC_SCAN_POINT;

   uint32_t maxStreams = options && options->max_streams ? options->max_streams : 1024;
   uint64_t sinkBytes = options && options->frame_sink_bytes ? options->frame_sink_bytes : (64ull << 20);

   maxStreams = (maxStreams + NFC_LANES - 1) / NFC_LANES * NFC_LANES;

   if (sinkBytes < 4ull * 4 * NFC_FRAME_MAX_WORDS)
      sinkBytes = 4ull * 4 * NFC_FRAME_MAX_WORDS;
   if (sinkBytes > (0xFFFFFFF0ull * 4ull))
      sinkBytes = 0xFFFFFFF0ull * 4ull;

   ctx->maxStreams = maxStreams;
   ctx->blocks = maxStreams / NFC_LANES;
   ctx->sinkWords = sinkBytes / 4;

   bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
   ok = ok && hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) == hipSuccess;
   {
      /* (the stream of the walk that writes the planes beside the rounds of second walks: whatever those rounds launch goes first) */
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess)
         least = 0;
      ok = ok && hipStreamCreateWithPriority(&ctx->low, hipStreamNonBlocking, least) == hipSuccess;
   }
   ok = ok && hipEventCreateWithFlags(&ctx->forkEvent, hipEventDisableTiming) == hipSuccess;
   ok = ok && hipEventCreateWithFlags(&ctx->joinEvent, hipEventDisableTiming) == hipSuccess;

   ok = ok && hipMalloc((void **)&ctx->dStates, sizeof(NfcStreamState) * (size_t)maxStreams) == hipSuccess;
   ok = ok && hipMalloc((void **)&ctx->dCold, sizeof(NfcStreamCold) * (size_t)maxStreams) == hipSuccess;
   ok = ok && hipMalloc((void **)&ctx->dRings, sizeof(float) * (size_t)kRingBlockFloats * ctx->blocks) == hipSuccess;
   ok = ok && hipMalloc((void **)&ctx->dBytes, (size_t)NFC_STREAM_BYTES * maxStreams) == hipSuccess;
   ok = ok && hipMalloc((void **)&ctx->dSink, ctx->sinkWords * 4) == hipSuccess;
   ok = ok && hipMalloc((void **)&ctx->dSinkCtl, 16) == hipSuccess;
   ok = ok && hipMalloc((void **)&ctx->dWorks, sizeof(NfcWork) * (size_t)maxStreams) == hipSuccess;
   ok = ok && hipMalloc((void **)&ctx->dConfigs, sizeof(NfcConfig) * kMaxConfigs) == hipSuccess;

   if (ok)
   {
      ok = ok && hipMemsetAsync(ctx->dStates, 0, sizeof(NfcStreamState) * (size_t)maxStreams, ctx->stream) == hipSuccess;
      ok = ok && hipMemsetAsync(ctx->dCold, 0, sizeof(NfcStreamCold) * (size_t)maxStreams, ctx->stream) == hipSuccess;
      ok = ok && hipMemsetAsync(ctx->dRings, 0, sizeof(float) * (size_t)kRingBlockFloats * ctx->blocks, ctx->stream) == hipSuccess;
      ok = ok && hipMemsetAsync(ctx->dBytes, 0, (size_t)NFC_STREAM_BYTES * maxStreams, ctx->stream) == hipSuccess;
      ok = ok && hipMemsetAsync(ctx->dSinkCtl, 0, 16, ctx->stream) == hipSuccess;
      ok = ok && hipStreamSynchronize(ctx->stream) == hipSuccess;
   }

   if (!ok)
   {
      nfcgpu_shutdown(ctx);
      return NFCGPU_ENOMEM;
   }

   ctx->ownSink = ctx->dSink;
   ctx->ownSinkCtl = ctx->dSinkCtl;
   ctx->ownSinkWords = ctx->sinkWords;

   ctx->streams.resize(maxStreams);
   ctx->hWorks.resize(maxStreams);
   for (auto &w: ctx->hWorks)
   {
      w.data = nullptr;
      w.count = 0;
      w.stride = 1;
   }

   *out = ctx;
   return NFCGPU_OK;
}

int nfcgpu_shutdown(nfcgpu_ctx *ctx)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   (void)hipSetDevice(ctx->device);

   if (ctx->stream)
      (void)hipStreamSynchronize(ctx->stream);
   if (ctx->side)
      (void)hipStreamSynchronize(ctx->side); /* (nothing of its work may outlive the buffers freed below) */

   (void)hipFree(ctx->dStates);
   (void)hipFree(ctx->dCold);
   (void)hipFree(ctx->dRings);
   (void)hipFree(ctx->dBytes);
   (void)hipFree(ctx->ownSink ? ctx->ownSink : ctx->dSink);
   (void)hipFree(ctx->ownSinkCtl ? ctx->ownSinkCtl : ctx->dSinkCtl);
   (void)hipFree(ctx->dWorks);
   (void)hipFree(ctx->dConfigs);
   for (nfcgpu_ctx::StageSlot &slot: ctx->stage)
   {
      if (slot.d)
         (void)hipFree(slot.d);
      if (slot.h)
         (void)hipHostFree(slot.h);
      if (slot.done)
         (void)hipEventDestroy(slot.done);
   }

   nfcgpu_comm_destroy(ctx);

   release_workspace(ctx);

   delete ctx;
   return NFCGPU_OK;
}

int nfcgpu_stream_open_many(nfcgpu_ctx *ctx, const nfcgpu_params *params, uint32_t count, uint32_t *first)
{
   if (!ctx || !first || count == 0)
      return NFCGPU_EINVAL;


   /* first fit of `count` contiguous free slots */
   uint32_t run = 0, start = 0;
   bool found = false;

   for (uint32_t i = 0; i < ctx->maxStreams; i++)
   {
      if (ctx->streams[i].open)
      {
         run = 0;
         continue;
      }
      if (run == 0)
         start = i;
      if (++run == count)
      {
         found = true;
         break;
      }
   }

   if (!found)
      return fail(ctx, NFCGPU_EFULL, "no free stream slots (raise nfcgpu_options.max_streams)");

   nfcgpu_params p;
   if (params)
      p = *params;
   else
      nfcgpu_default_params(&p);

   for (uint32_t i = start; i < start + count; i++)
   {
      StreamInfo &si = ctx->streams[i];
      si = StreamInfo();
      si.open = true;
      si.params = p;
      si.powerAtInit = p.power_level_threshold;
   }

   *first = start;
   return NFCGPU_OK;
}

int nfcgpu_stream_open(nfcgpu_ctx *ctx, const nfcgpu_params *params, uint32_t *id)
{
   return nfcgpu_stream_open_many(ctx, params, 1, id);
}

int nfcgpu_stream_configure(nfcgpu_ctx *ctx, uint32_t id, const nfcgpu_params *params)
{
   if (!ctx || !params)
      return NFCGPU_EINVAL;

   if (id >= ctx->maxStreams || !ctx->streams[id].open)
      return fail(ctx, NFCGPU_ESTREAM, "unknown stream");

   StreamInfo &si = ctx->streams[id];
   const uint32_t oldRate = si.params.sample_rate;

   if (params->sample_rate != 0 && params->sample_rate != oldRate && oldRate != 0 && ctx->dirty && !ctx->hold)
   {
      /* frames already produced keep the rate stored when they were produced */
      int rc = nfcgpu_sync(ctx);
      if (rc != NFCGPU_OK && rc != NFCGPU_EOVERFLOW)
         return rc;
   }

   si.params = *params;

   /* setSampleRate() only stores the value (0 = leave it alone); nothing is re-derived from it until the decoder
    * (re)initialises, see adopt_sample_rate */
   if (params->sample_rate == 0)
      si.params.sample_rate = oldRate;

   if (si.initialized && !si.needInit)
      return resolve_config(ctx, si); /* thresholds and the enable mask take effect immediately */

   return NFCGPU_OK;
}

int nfcgpu_stream_reset(nfcgpu_ctx *ctx, uint32_t id)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   if (id >= ctx->maxStreams || !ctx->streams[id].open)
      return fail(ctx, NFCGPU_ESTREAM, "unknown stream");

   /* initialize(): everything is derived again from what is stored now (applied when the next buffer arrives) */
   StreamInfo &si = ctx->streams[id];
   si.needInit = true;
   si.explicitInit = true;
   si.derivedRate = si.params.sample_rate;
   si.powerAtInit = si.params.power_level_threshold;
   return NFCGPU_OK;
}

int nfcgpu_stream_close(nfcgpu_ctx *ctx, uint32_t id)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   if (id >= ctx->maxStreams || !ctx->streams[id].open)
      return fail(ctx, NFCGPU_ESTREAM, "unknown stream");

   int rc = nfcgpu_sync(ctx);

   ctx->streams[id] = StreamInfo();
   return rc;
}

int nfcgpu_submit_batch(nfcgpu_ctx *ctx, const nfcgpu_batch *b)
{
   if (!ctx || !b || !b->stream_ids || !b->data || !b->n_samples || (b->stride != 1 && b->stride != 2) ||
       (b->location != NFCGPU_LOC_HOST && b->location != NFCGPU_LOC_DEVICE))
      return NFCGPU_EINVAL;

   if (b->n_streams == 0)
      return NFCGPU_OK;

   HIP_TRY(ctx, hipSetDevice(ctx->device));

   /* the per-slot work table and the batch marks are idle between calls: every exit path restores that */
   auto clearWorks = [&]() {
      for (uint32_t i = 0; i < b->n_streams; i++)
      {
         const uint32_t id = b->stream_ids[i];
         if (id >= ctx->maxStreams)
            continue;
         NfcWork &w = ctx->hWorks[id];
         w.data = nullptr;
         w.count = 0;
         w.stride = 1;
         ctx->streams[id].listed = false;
      }
   };

   /* validate + adopt rate */
   size_t hostBytes = 0;
   uint32_t lo = 0xFFFFFFFFu, hi = 0;
   int rc = NFCGPU_OK;

   for (uint32_t i = 0; i < b->n_streams && rc == NFCGPU_OK; i++)
   {
      const uint32_t id = b->stream_ids[i];

      if (id >= ctx->maxStreams || !ctx->streams[id].open)
         rc = fail(ctx, NFCGPU_ESTREAM, "unknown stream in batch");
      else if (ctx->streams[id].listed)
         rc = fail(ctx, NFCGPU_EINVAL, "stream listed twice in one batch");
      else if (b->n_samples[i] && !b->data[i])
         rc = fail(ctx, NFCGPU_EINVAL, "null data pointer in batch");
      else
         rc = adopt_sample_rate(ctx, ctx->streams[id], b->sample_rate);

      if (rc)
         break;

      ctx->streams[id].listed = true;
      ctx->hWorks[id].count = b->n_samples[i];
      hostBytes += (size_t)b->n_samples[i] * b->stride * 4;
      lo = id < lo ? id : lo;
      hi = id > hi ? id : hi;
   }

   if (rc)
   {
      clearWorks();
      return rc;
   }

   /* one launch per decoder configuration present in the batch */
   std::vector<uint32_t> cfgs;
   for (uint32_t i = 0; i < b->n_streams; i++)
   {
      uint32_t c = ctx->streams[b->stream_ids[i]].config;
      bool seen = false;
      for (uint32_t k: cfgs)
         seen = seen || k == c;
      if (!seen)
         cfgs.push_back(c);
   }

   /* staging slot: the samples (host batches) and one work table per configuration group, all sent from pinned memory
    * so that nothing the copies read belongs to this call's stack or to the caller once it returns */
   const size_t tableBytes = (sizeof(NfcWork) * (size_t)(hi - lo + 1) + 255) & ~(size_t)255;
   nfcgpu_ctx::StageSlot *slot = nullptr;

   rc = stage_acquire(ctx, (b->location == NFCGPU_LOC_HOST ? hostBytes + 256 * (size_t)b->n_streams : 0) + tableBytes * (cfgs.size() + 1) + 256, &slot);
   if (rc)
   {
      clearWorks();
      return rc;
   }

   ctx->inflight = true;

   /* from here on every exit records the slot's event behind whatever has been enqueued */
   struct Release
   {
      nfcgpu_ctx *ctx;
      nfcgpu_ctx::StageSlot *slot;
      ~Release() { stage_release(ctx, slot); }
   } release {ctx, slot};

   rc = initialize_pending(ctx, lo, hi - lo + 1, true);
   if (rc)
   {
      clearWorks();
      return rc;
   }

   size_t stageAt = 0;

   for (uint32_t i = 0; i < b->n_streams; i++)
   {
      const uint32_t id = b->stream_ids[i];
      NfcWork &w = ctx->hWorks[id];
      const size_t bytes = (size_t)b->n_samples[i] * b->stride * 4;

      w.stride = b->stride;

      if (b->location == NFCGPU_LOC_HOST)
      {
         w.data = slot->d + stageAt;
         if (bytes)
            std::memcpy(slot->h + stageAt, b->data[i], bytes);
         stageAt += (bytes + 255) & ~(size_t)255;
      }
      else
      {
         w.data = (const uint8_t *)b->data[i];
      }
   }

   if (stageAt)
   {
      hipError_t err = hipMemcpyAsync(slot->d, slot->h, stageAt, hipMemcpyHostToDevice, ctx->stream);
      if (err != hipSuccess)
      {
         clearWorks();
         return fail(ctx, NFCGPU_EHIP, "hipMemcpyAsync(H2D samples)", err);
      }
   }

   size_t tableAt = (stageAt + 255) & ~(size_t)255; /* the groups' work tables follow the samples in the pinned mirror */

   for (uint32_t c: cfgs)
   {
      {
         std::vector<WindowedItem> items;
         for (uint32_t i = 0; i < b->n_streams; i++)
         {
            const uint32_t id = b->stream_ids[i];
            if (ctx->streams[id].config == c)
               items.push_back(WindowedItem {id, ctx->hWorks[id].data, b->n_samples[i]});
         }

         if (windowed_eligible(ctx, c, items))
         {
            rc = run_windowed(ctx, c, items, b->stride);
            if (rc)
            {
               clearWorks();
               return rc;
            }
            continue;
         }
      }

      uint32_t first = 0xFFFFFFFFu, last = 0;
      uint64_t groupSamples = 0;
      bool exactPossible = false, exactOnly = true;

      for (uint32_t i = 0; i < b->n_streams; i++)
      {
         const uint32_t id = b->stream_ids[i];
         if (ctx->streams[id].config != c)
            continue;
         first = id < first ? id : first;
         last = id > last ? id : last;
         groupSamples += b->n_samples[i];
         const bool exact = exact_zone(ctx->streams[id].clock, b->n_samples[i]);
         exactPossible = exactPossible || exact;
         exactOnly = exactOnly && (exact || b->n_samples[i] == 0);
      }

      /* slots of other configurations inside [first,last] must stay idle in this launch */
      NfcWork *table = (NfcWork *)(slot->h + tableAt);
      const size_t entries = (size_t)last - first + 1;
      tableAt += tableBytes;

      for (uint32_t id = first; id <= last; id++)
      {
         table[id - first] = ctx->hWorks[id];

         if (!ctx->streams[id].open || ctx->streams[id].config != c)
         {
            table[id - first].data = nullptr;
            table[id - first].count = 0;
            table[id - first].stride = 1;
         }
      }

      /* (stream order: the copy lands after the launch of the group before has finished with the device table) */
      hipError_t err = hipMemcpyAsync(ctx->dWorks + first, table, sizeof(NfcWork) * entries, hipMemcpyHostToDevice, ctx->stream);
      if (err != hipSuccess)
      {
         clearWorks();
         return fail(ctx, NFCGPU_EHIP, "hipMemcpyAsync(work table)", err);
      }

      NfcLaunch L = base_launch(ctx);
      L.works = ctx->dWorks;
      L.uniformStride = b->stride; /* one sample format per batch */
      L.firstSlot = first;
      L.slotCount = last - first + 1;

      rc = launch_demod(ctx, c, L, groupSamples, exactPossible, exactOnly);
      if (rc)
      {
         clearWorks();
         return rc;
      }

      /* the clock mirrors follow once the launch is on its way */
      for (uint32_t i = 0; i < b->n_streams; i++)
      {
         const uint32_t id = b->stream_ids[i];
         if (ctx->streams[id].config == c)
            commit_clock(ctx->streams[id], b->n_samples[i]);
      }
   }

   /* no wait: the staging slot and its tables are protected by the slot's event (stage_release), the results are
    * collected by whoever asks for them (nfcgpu_sync / poll / flush) */
   clearWorks();
   return NFCGPU_OK;
}

int nfcgpu_submit(nfcgpu_ctx *ctx, uint32_t id, const float *data, uint32_t n, uint32_t stride, uint32_t sampleRate)
{
   const void *ptr = data;
   nfcgpu_batch b;
   std::memset(&b, 0, sizeof(b));
   b.n_streams = 1;
   b.stride = stride;
   b.location = NFCGPU_LOC_HOST;
   b.sample_rate = sampleRate;
   b.stream_ids = &id;
   b.data = &ptr;
   b.n_samples = &n;
   return nfcgpu_submit_batch(ctx, &b);
}

int nfcgpu_magnitude(nfcgpu_ctx *ctx, const float *iq, uint64_t n, float *out, uint32_t location)
{
   if (!ctx || !iq || !out || (location != NFCGPU_LOC_HOST && location != NFCGPU_LOC_DEVICE))
      return NFCGPU_EINVAL;
   if (n == 0)
      return NFCGPU_OK;

   HIP_TRY(ctx, hipSetDevice(ctx->device));

   const float2 *src = (const float2 *)iq;
   float *dst = out;
   nfcgpu_ctx::StageSlot *slot = nullptr;

   ctx->inflight = true;

   if (location == NFCGPU_LOC_HOST)
   {
      /* staging area: IQ first, magnitudes behind it */
      const size_t inBytes = (size_t)n * 8, outBytes = (size_t)n * 4;
      int rc = stage_acquire(ctx, inBytes + outBytes, &slot);
      if (rc)
         return rc;

      std::memcpy(slot->h, iq, inBytes);
      HIP_TRY(ctx, hipMemcpyAsync(slot->d, slot->h, inBytes, hipMemcpyHostToDevice, ctx->stream));
      src = (const float2 *)slot->d;
      dst = (float *)(slot->d + inBytes);
   }

   const uint32_t threads = 256;
   const uint64_t wanted = (n + threads - 1) / threads;
   const uint32_t grid = (uint32_t)(wanted < 16384 ? wanted : 16384);

   hipLaunchKernelGGL(nfc_magnitude_kernel, dim3(grid), dim3(threads), 0, ctx->stream, src, dst, n);
   HIP_TRY(ctx, hipGetLastError());

   if (location == NFCGPU_LOC_HOST)
      HIP_TRY(ctx, hipMemcpyAsync(slot->h + (size_t)n * 8, dst, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));

   /* the result goes back to the caller: this entry point waits */
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

   if (location == NFCGPU_LOC_HOST)
      std::memcpy(out, slot->h + (size_t)n * 8, (size_t)n * 4);

   return NFCGPU_OK;
}

int nfcgpu_resample_radio(nfcgpu_ctx *ctx, const float *in, uint64_t inPitch, uint32_t nBuffers, uint32_t n, float *out, uint64_t outPitch,
                          uint32_t capacityPairs, uint32_t *counts, uint32_t location)
{
   if (!ctx || !in || !out || !counts || n < 25 || (inPitch & 3) || (outPitch & 7) || ((uintptr_t)out & 7) || inPitch < (uint64_t)n * 4 ||
       outPitch < (uint64_t)capacityPairs * 8 || (location != NFCGPU_LOC_HOST && location != NFCGPU_LOC_DEVICE))
      return NFCGPU_EINVAL;
   if (nBuffers == 0)
      return NFCGPU_OK;

   HIP_TRY(ctx, hipSetDevice(ctx->device));

   const float *dIn = in;
   float *dOut = out;
   uint32_t *dCounts = counts;
   uint8_t *scratch = nullptr;
   std::vector<uint32_t> hostCounts;

   const size_t inBytes = (size_t)inPitch * nBuffers, outBytes = (size_t)outPitch * nBuffers, cntBytes = (size_t)nBuffers * 4;

   if (location == NFCGPU_LOC_HOST)
   {
      /* one temporary device block: input, output, counts (this entry point is not on the streaming path) */
      if (hipMalloc((void **)&scratch, inBytes + outBytes + cntBytes) != hipSuccess)
         return fail(ctx, NFCGPU_ENOMEM, "resampler scratch allocation failed");

      hipError_t err = hipMemcpy(scratch, in, inBytes, hipMemcpyHostToDevice);
      if (err != hipSuccess)
      {
         (void)hipFree(scratch);
         return fail(ctx, NFCGPU_EHIP, "hipMemcpy(H2D resampler input)", err);
      }

      dIn = (const float *)scratch;
      dOut = (float *)(scratch + inBytes);
      dCounts = (uint32_t *)(scratch + inBytes + outBytes);
   }

   hipLaunchKernelGGL(nfc_resample_radio_kernel, dim3((nBuffers + NFC_LANES - 1) / NFC_LANES), dim3(NFC_LANES), 0, ctx->stream, dIn,
                      inPitch / 4, nBuffers, n, dOut, outPitch / 4, capacityPairs, dCounts);

   hipError_t err = hipGetLastError();
   if (err == hipSuccess)
      err = hipStreamSynchronize(ctx->stream);

   hostCounts.resize(nBuffers);

   if (err == hipSuccess)
      err = hipMemcpy(hostCounts.data(), dCounts, cntBytes, hipMemcpyDeviceToHost);

   if (err == hipSuccess && location == NFCGPU_LOC_HOST)
   {
      err = hipMemcpy(out, dOut, outBytes, hipMemcpyDeviceToHost);
      std::memcpy(counts, hostCounts.data(), cntBytes);
   }

   if (scratch)
      (void)hipFree(scratch);

   if (err != hipSuccess)
      return fail(ctx, NFCGPU_EHIP, "adaptive resampler", err);

   for (uint32_t b = 0; b < nBuffers; b++)
   {
      if (hostCounts[b] > capacityPairs)
         return fail(ctx, NFCGPU_EOVERFLOW, "resampler output capacity exceeded: raise capacity_pairs");
   }

   return NFCGPU_OK;
}

int nfcgpu_submit_uniform(nfcgpu_ctx *ctx, uint32_t first, uint32_t count, const void *base, uint64_t pitch, uint32_t n,
                          uint32_t stride, uint32_t location, uint32_t sampleRate)
{
   if (!ctx || !base || (stride != 1 && stride != 2) || count == 0 ||
       (location != NFCGPU_LOC_HOST && location != NFCGPU_LOC_DEVICE) || (count > 1 && pitch < (uint64_t)n * stride * 4))
      return NFCGPU_EINVAL;
   if ((uint64_t)first + count > ctx->maxStreams)
      return fail(ctx, NFCGPU_ESTREAM, "stream range out of bounds");

   /* rows are read as float2 (IQ) or float: base and pitch have to be aligned to a sample */
   if (((uintptr_t)base % (4u * stride)) != 0 || (count > 1 && (pitch % (4u * stride)) != 0))
      return fail(ctx, NFCGPU_EINVAL, "base and pitch must be multiples of the sample size (4 bytes magnitude, 8 bytes IQ)");

   HIP_TRY(ctx, hipSetDevice(ctx->device));

   /* an empty buffer still stores a new sample rate and re-initialises the stream, like nfcgpu_submit (NfcDecoder.cpp:383-388) */
   for (uint32_t i = first; i < first + count; i++)
   {
      if (!ctx->streams[i].open)
         return fail(ctx, NFCGPU_ESTREAM, "closed stream inside uniform range");

      int rc = adopt_sample_rate(ctx, ctx->streams[i], sampleRate);
      if (rc)
         return rc;
   }

   if (n == 0)
      return NFCGPU_OK;

   const uint8_t *devBase = (const uint8_t *)base;
   uint64_t devPitch = pitch;
   nfcgpu_ctx::StageSlot *slot = nullptr;

   ctx->inflight = true;

   /* host rows go through a staging slot; its event is recorded behind whatever this call enqueues, on every exit */
   struct Release
   {
      nfcgpu_ctx *ctx;
      nfcgpu_ctx::StageSlot *slot;
      ~Release() { if (slot) stage_release(ctx, slot); }
   } release {ctx, nullptr};

   if (location == NFCGPU_LOC_HOST)
   {
      const size_t row = (size_t)n * stride * 4;
      devPitch = (row + 255) & ~(size_t)255;

      int rc = stage_acquire(ctx, devPitch * count, &slot);
      if (rc)
         return rc;

      for (uint32_t r = 0; r < count; r++)
         std::memcpy(slot->h + (size_t)r * devPitch, (const uint8_t *)base + (size_t)r * pitch, row);

      release.slot = slot;

      HIP_TRY(ctx, hipMemcpyAsync(slot->d, slot->h, devPitch * count, hipMemcpyHostToDevice, ctx->stream));
      devBase = slot->d;
   }

   int rc = initialize_pending(ctx, first, count, false);
   if (rc)
      return rc;

   rc = run_rows(ctx, first, count, devBase, devPitch, n, stride);
   if (rc)
      return rc;

   /* host buffers are never retained past the call: they were copied into the staging slot */
   return NFCGPU_OK;
}

int nfcgpu_sync(nfcgpu_ctx *ctx)
{
   if (!ctx)
      return NFCGPU_EINVAL;


   /* nothing enqueued since the last synchronisation and nothing to collect: no device call at all */
   if (!ctx->inflight && !ctx->dirty && ctx->timed.empty() && ctx->timedScan.empty() && ctx->timedWindow.empty() && ctx->timedWave.empty() && ctx->timedPlanes.empty())
      return NFCGPU_OK;

   HIP_TRY(ctx, hipSetDevice(ctx->device));
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
   ctx->inflight = false;

   if (!ctx->timedWave.empty() || !ctx->timedPlanes.empty())
      (void)hipStreamSynchronize(ctx->side); /* (carry lanes run beside the windows) */

   collect_timings(ctx);

   if (!ctx->dirty || ctx->hold)
      return NFCGPU_OK;

   uint32_t ctl[2] = {0, 0};
   HIP_TRY(ctx, hipMemcpy(ctl, ctx->dSinkCtl, sizeof(ctl), hipMemcpyDeviceToHost));

   const uint64_t used = ctl[0] < ctx->sinkWords ? ctl[0] : ctx->sinkWords;

   if (used)
   {
      ctx->hSink.resize(used);
      HIP_TRY(ctx, hipMemcpy(ctx->hSink.data(), ctx->dSink, used * 4, hipMemcpyDeviceToHost));
      drain_sink(ctx, ctl[0]);
   }

   ctx->stats.dropped_frames += ctl[1];

   HIP_TRY(ctx, hipMemsetAsync(ctx->dSinkCtl, 0, 16, ctx->stream));
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

   ctx->dirty = false;

   if (ctl[1])
      return fail(ctx, NFCGPU_EOVERFLOW, "frame sink overflow: frames were dropped (raise frame_sink_bytes or sync more often)");

   return NFCGPU_OK;
}

int nfcgpu_flush(nfcgpu_ctx *ctx, uint32_t id)
{
   if (!ctx)
      return NFCGPU_EINVAL;
   if (id >= ctx->maxStreams || !ctx->streams[id].open)
      return fail(ctx, NFCGPU_ESTREAM, "unknown stream");

   int rc = nfcgpu_sync(ctx);
   if (rc && rc != NFCGPU_EOVERFLOW)
      return rc;

   StreamInfo &si = ctx->streams[id];

   uint32_t clock = 0xFFFFFFFFu, carrierOn = 0;

   if (si.initialized)
   {
      NfcStreamState s;
      HIP_TRY(ctx, hipMemcpy(&s, ctx->dStates + id, sizeof(s), hipMemcpyDeviceToHost));
      clock = si.needInit ? 0xFFFFFFFFu : s.clock; /* a pending initialize() has already reset the clock, not the carrier state */
      carrierOn = s.carrierOn;
   }

   nfcgpu_frame f;
   std::memset(&f, 0, sizeof(f));
   f.stream_id = id;
   f.tech_type = NFC_TECH_ANY;
   f.frame_type = carrierOn ? NFC_FRAME_CARRIER_ON : NFC_FRAME_CARRIER_OFF;
   f.frame_phase = NFC_PHASE_CARRIER;
   f.sample_start = clock;
   f.sample_end = clock;
   f.sample_rate = si.params.sample_rate;

   si.queue.push_back(f);
   return rc;
}

int nfcgpu_poll(nfcgpu_ctx *ctx, uint32_t id, nfcgpu_frame *out, uint32_t capacity, uint32_t *count)
{
   if (!ctx || !count || (capacity && !out))
      return NFCGPU_EINVAL;
   if (id >= ctx->maxStreams || !ctx->streams[id].open)
      return fail(ctx, NFCGPU_ESTREAM, "unknown stream");

   int rc = nfcgpu_sync(ctx);
   if (rc && rc != NFCGPU_EOVERFLOW)
      return rc;

   StreamInfo &si = ctx->streams[id];
   uint32_t n = 0;

   while (n < capacity && !si.queue.empty())
   {
      out[n++] = si.queue.front();
      si.queue.pop_front();
   }

   *count = n;
   return rc;
}

/* the frames the device has decoded for a stream, as the trace file the reference application opens
 * (TraceStorageTask.cpp:211-240 the range of "write file", 461-520 the entries). The frames stay in the stream's queue. */
int nfcgpu_trace_write(nfcgpu_ctx *ctx, uint32_t id, const char *path, double rangeStart, double rangeEnd, uint32_t *written)
{
   if (!ctx || !path)
      return NFCGPU_EINVAL;
   if (id >= ctx->maxStreams || !ctx->streams[id].open)
      return fail(ctx, NFCGPU_ESTREAM, "unknown stream");

   /* (a held sink is not drained by nfcgpu_sync: the stream's queue would be empty and the file with it) */
   if (ctx->hold)
      return fail(ctx, NFCGPU_EINVAL, "nfcgpu_trace_write while the sink is held (nfcgpu_sink_hold): the frames are in the caller's sink, not in the stream's queue");

   int rc = nfcgpu_sync(ctx);
   if (rc && rc != NFCGPU_EOVERFLOW)
      return rc;

   const StreamInfo &si = ctx->streams[id];
   std::vector<nfcgpu_frame> frames(si.queue.begin(), si.queue.end());

   const int wrote = nfcgpu_trace_write_frames(path, frames.data(), (uint32_t)frames.size(), si.params.stream_time, rangeStart, rangeEnd, written);
   if (wrote)
      return fail(ctx, wrote, "the trace file could not be written");

   return rc;
}

int nfcgpu_pending(nfcgpu_ctx *ctx, uint32_t id, uint32_t *count)
{
   if (!ctx || !count)
      return NFCGPU_EINVAL;
   if (id >= ctx->maxStreams || !ctx->streams[id].open)
      return fail(ctx, NFCGPU_ESTREAM, "unknown stream");

   int rc = nfcgpu_sync(ctx);
   *count = (uint32_t)ctx->streams[id].queue.size();
   return rc;
}

int nfcgpu_sink_device_view(nfcgpu_ctx *ctx, const void **words, const void **cursor, uint64_t *capacity)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   if (words)
      *words = ctx->dSink;
   if (cursor)
      *cursor = ctx->dSinkCtl;
   if (capacity)
      *capacity = ctx->sinkWords;
   return NFCGPU_OK;
}

int nfcgpu_sink_attach(nfcgpu_ctx *ctx, void *words, uint64_t capacityWords, void *ctl)
{
   if (!ctx || (words && (!ctl || capacityWords < 4ull * NFC_FRAME_MAX_WORDS || capacityWords > 0xFFFFFFF0ull)))
      return NFCGPU_EINVAL;


   const bool wasHeld = ctx->hold;
   ctx->hold = false;
   int rc = nfcgpu_sync(ctx); /* drain what the current sink holds */
   ctx->hold = wasHeld;

   if (rc && rc != NFCGPU_EOVERFLOW)
      return rc;

   if (words)
   {
      ctx->dSink = (uint32_t *)words;
      ctx->dSinkCtl = (uint32_t *)ctl;
      ctx->sinkWords = capacityWords;
   }
   else
   {
      ctx->dSink = ctx->ownSink;
      ctx->dSinkCtl = ctx->ownSinkCtl;
      ctx->sinkWords = ctx->ownSinkWords;
   }

   return nfcgpu_sink_rewind(ctx);
}

int nfcgpu_sink_hold(nfcgpu_ctx *ctx, int hold)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   ctx->hold = hold != 0;
   return NFCGPU_OK;
}

int nfcgpu_sink_rewind(nfcgpu_ctx *ctx)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   HIP_TRY(ctx, hipSetDevice(ctx->device));
   HIP_TRY(ctx, hipMemsetAsync(ctx->dSinkCtl, 0, 16, ctx->stream));
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
   ctx->dirty = false;
   return NFCGPU_OK;
}


int nfcgpu_comm_unique_id(void *id128)
{
   if (!id128)
      return NFCGPU_EINVAL;
   Rccl *r = rccl();
   if (!r)
      return NFCGPU_ENODEV;
   return r->getUniqueId(id128) == 0 ? NFCGPU_OK : NFCGPU_EHIP;
}

int nfcgpu_comm_init(nfcgpu_ctx *ctx, const void *id128, int rank, int nRanks)
{
   if (!ctx || !id128 || nRanks < 1 || rank < 0 || rank >= nRanks)
      return NFCGPU_EINVAL;

   Rccl *r = rccl();
   if (!r)
      return fail(ctx, NFCGPU_ENODEV, "librccl.so not found");

   HIP_TRY(ctx, hipSetDevice(ctx->device));

   if (ctx->comm)
      nfcgpu_comm_destroy(ctx);

   IdByValue id;
   std::memcpy(id.internal, id128, NFCGPU_UNIQUE_ID_BYTES);

   const int rc = r->commInitRank(&ctx->comm, nRanks, id, rank);
   if (rc != 0)
   {
      ctx->comm = nullptr;
      return fail(ctx, NFCGPU_EHIP, r->getErrorString ? r->getErrorString(rc) : "ncclCommInitRank failed");
   }

   ctx->commRank = rank;
   ctx->commRanks = nRanks;

   if (hipMalloc((void **)&ctx->dCounts, 8 * (size_t)(nRanks + 1)) != hipSuccess)
      return fail(ctx, NFCGPU_ENOMEM, "hipMalloc(gather counts)");

   return NFCGPU_OK;
}

int nfcgpu_comm_destroy(nfcgpu_ctx *ctx)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   /* (librccl.so is only looked up for a context that has a communicator: a plain shutdown - possibly from an atexit
    * handler of the host - must not load a library) */
   Rccl *r = ctx->comm ? rccl() : nullptr;

   if (ctx->comm && r)
   {
      (void)hipSetDevice(ctx->device);
      (void)hipStreamSynchronize(ctx->stream);
      (void)r->commDestroy(ctx->comm);
   }

   ctx->comm = nullptr;
   ctx->commRanks = 0;

   if (ctx->dCounts)
      (void)hipFree(ctx->dCounts);
   ctx->dCounts = nullptr;

   return NFCGPU_OK;
}

/* both forms of the gather: `packed` - rank r's records start at the sum of the counts before it; otherwise at r * stride,
 * stride = the largest count (the layout of rounds 1-2, kept under the old symbol) */
static int gather_frames(nfcgpu_ctx *ctx, void *gathered, uint64_t capacityWords, uint32_t *countsHost, uint64_t *strideWords, bool packed)
{
   if (!ctx || !gathered || !countsHost || (!packed && !strideWords))
      return NFCGPU_EINVAL;

   if (!ctx->comm)
      return fail(ctx, NFCGPU_EINVAL, "nfcgpu_comm_init first");
   if (!ctx->hold)
      return fail(ctx, NFCGPU_EINVAL, "nfcgpu_sink_hold first: a sink that nfcgpu_sync drains has nothing left to gather");

   Rccl *r = rccl();
   if (!r)
      return fail(ctx, NFCGPU_ENODEV, "librccl.so not found");

   HIP_TRY(ctx, hipSetDevice(ctx->device));
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

   /* this rank's records: what the sink holds, clamped like drain_sink does (a record is only ever written below
    * sinkWords - NFC_FRAME_MAX_WORDS) */
   uint32_t ctl[2] = {0, 0};
   HIP_TRY(ctx, hipMemcpy(ctl, ctx->dSinkCtl, sizeof(ctl), hipMemcpyDeviceToHost));

   uint64_t used = ctl[0];
   const uint64_t limit = ctx->sinkWords >= NFC_FRAME_MAX_WORDS ? ctx->sinkWords - NFC_FRAME_MAX_WORDS + 1 : 0;
   if (ctl[1] && used > limit)
      used = limit; /* dropped frames: everything that starts below the limit is whole (nfc_emit) */
   if (used > ctx->sinkWords)
      used = ctx->sinkWords;

   const int n = ctx->commRanks;

   /* Every rank learns every rank's word count AND what every rank's receive buffer holds: the decision to go on is
    * then the same on all of them (a rank that returned between the two collectives would leave the others waiting). */
   const uint32_t mine[2] = {(uint32_t)used, (uint32_t)(capacityWords > 0xFFFFFFFFull ? 0xFFFFFFFFull : capacityWords)};

   HIP_TRY(ctx, hipMemcpyAsync(ctx->dCounts + 2 * n, mine, 8, hipMemcpyHostToDevice, ctx->stream));
   int rc = r->allGather(ctx->dCounts + 2 * n, ctx->dCounts, 2, /* ncclUint32 */ 3, ctx->comm, ctx->stream);
   if (rc != 0)
      return fail(ctx, NFCGPU_EHIP, r->getErrorString ? r->getErrorString(rc) : "ncclAllGather(counts) failed");

   std::vector<uint32_t> pairs(2 * (size_t)n);
   HIP_TRY(ctx, hipMemcpyAsync(pairs.data(), ctx->dCounts, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

   uint64_t total = 0, smallest = ~0ull, largest = 0;
   for (int i = 0; i < n; i++)
   {
      countsHost[i] = pairs[2 * i];
      total += pairs[2 * i];
      largest = pairs[2 * i] > largest ? pairs[2 * i] : largest;
      smallest = pairs[2 * i + 1] < smallest ? pairs[2 * i + 1] : smallest;
   }

   const uint64_t stride = packed ? 0 : largest;
   const uint64_t needed = packed ? total : largest * (uint64_t)n;

   if (strideWords)
      *strideWords = stride;

   if (needed > smallest)
      return fail(ctx, NFCGPU_ENOMEM, "a rank's gather buffer is too small for all ranks' records (the same verdict on every rank)");

   /* records, exact sizes: one broadcast per rank into its place, all in one group */
   rc = r->groupStart();
   uint64_t at = 0;
   for (int i = 0; i < n && rc == 0; i++)
   {
      if (countsHost[i])
         rc = r->broadcast(ctx->dSink, (uint32_t *)gathered + (packed ? at : (uint64_t)i * stride), countsHost[i], /* ncclUint32 */ 3, i, ctx->comm, ctx->stream);
      at += countsHost[i];
   }
   const int rcEnd = r->groupEnd();
   if (rc == 0)
      rc = rcEnd;
   if (rc != 0)
      return fail(ctx, NFCGPU_EHIP, r->getErrorString ? r->getErrorString(rc) : "ncclBroadcast(records) failed");

   HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

   return ctl[1] ? NFCGPU_EOVERFLOW : NFCGPU_OK;
}

int nfcgpu_gather_frames_packed(nfcgpu_ctx *ctx, void *gathered, uint64_t capacityWords, uint32_t *countsHost)
{
   return gather_frames(ctx, gathered, capacityWords, countsHost, nullptr, true);
}

int nfcgpu_gather_frames(nfcgpu_ctx *ctx, void *gathered, uint64_t capacityWords, uint32_t *countsHost, uint64_t *strideWords)
{
   return gather_frames(ctx, gathered, capacityWords, countsHost, strideWords, false);
}

int nfcgpu_read_bandwidth(nfcgpu_ctx *ctx, const void *ptr, uint64_t bytes, uint32_t repeats, double *gbps)
{
   if (!ctx || !ptr || bytes < 16 || !gbps || ((uintptr_t)ptr & 15))
      return NFCGPU_EINVAL;


   HIP_TRY(ctx, hipSetDevice(ctx->device));

   float *out = nullptr;
   HIP_TRY(ctx, hipMalloc((void **)&out, 16));

   hipEvent_t a = take_event(ctx), b = take_event(ctx);
   double best = 0.0;

   for (uint32_t i = 0; i < (repeats ? repeats : 1); i++)
   {
      (void)hipEventRecord(a, ctx->stream);
      hipLaunchKernelGGL(nfc_read_kernel, dim3(256 * 16), dim3(256), 0, ctx->stream, (const float4 *)ptr, bytes / 16, out);
      (void)hipEventRecord(b, ctx->stream);
      (void)hipStreamSynchronize(ctx->stream);

      float ms = 0;
      if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0)
      {
         const double g = (double)(bytes / 16 * 16) / (ms * 1e-3) / 1e9;
         best = g > best ? g : best;
      }
   }

   ctx->eventPool.push_back(a);
   ctx->eventPool.push_back(b);
   (void)hipFree(out);

   *gbps = best;
   return NFCGPU_OK;
}

namespace {

/* total length of the union of the wave decoder's launch intervals */
void close_wave_spans(nfcgpu_ctx *ctx)
{
   std::vector<std::pair<float, float>> &v = ctx->waveSpans;
   std::sort(v.begin(), v.end());

   double busy = ctx->waveBusyMs; /* (what fold_wave_spans has taken out of the list) */
   float lo = 0, hi = -1.0f;

   for (const auto &span: v)
   {
      if (hi < lo || span.first > hi)
      {
         if (hi >= lo)
            busy += hi - lo;
         lo = span.first;
         hi = span.second;
      }
      else if (span.second > hi)
         hi = span.second;
   }

   if (hi >= lo)
      busy += hi - lo;

   ctx->stats.wave_busy_ms = busy;
}

}

int nfcgpu_stats_get(nfcgpu_ctx *ctx, nfcgpu_stats *stats)
{
   if (!ctx || !stats)
      return NFCGPU_EINVAL;

   std::memcpy(stats, &ctx->stats, NFCGPU_STATS_SIZE_V2); /* a caller built against the older header holds no more */
   return NFCGPU_OK;
}

int nfcgpu_stats_get_sized(nfcgpu_ctx *ctx, void *stats, uint32_t size)
{
   if (!ctx || !stats)
      return NFCGPU_EINVAL;

   close_wave_spans(ctx);

   std::memcpy(stats, &ctx->stats, size < sizeof(nfcgpu_stats) ? size : sizeof(nfcgpu_stats));
   return NFCGPU_OK;
}

int nfcgpu_stats_reset(nfcgpu_ctx *ctx)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   ctx->stats = nfcgpu_stats();
   ctx->waveSpans.clear();
   ctx->waveBusyMs = 0.0;

   /* the time base of the launch intervals: an event on the context's stream, now (on the context's device: with several
    * contexts in a process - a rank per GPU - another one may be current) */
   HIP_TRY(ctx, hipSetDevice(ctx->device));
   if (!ctx->epoch)
      HIP_TRY(ctx, hipEventCreate(&ctx->epoch));
   HIP_TRY(ctx, hipEventRecord(ctx->epoch, ctx->stream));

   return NFCGPU_OK;
}

int nfcgpu_profile(nfcgpu_ctx *ctx, int enable)
{
   if (!ctx)
      return NFCGPU_EINVAL;

   ctx->profile = enable != 0;
   return NFCGPU_OK;
}

void *nfcgpu_hip_stream(nfcgpu_ctx *ctx)
{
   return ctx ? (void *)ctx->stream : nullptr;
}

const char *nfcgpu_strerror(int code)
{
   switch (code)
   {
      case NFCGPU_OK: return "ok";
      case NFCGPU_EINVAL: return "invalid argument";
      case NFCGPU_ENODEV: return "no usable HIP device (this library has no CPU fallback)";
      case NFCGPU_ENOMEM: return "out of memory";
      case NFCGPU_ESTREAM: return "unknown or closed stream";
      case NFCGPU_ERATE: return "sample rate not decodable";
      case NFCGPU_EOVERFLOW: return "frame sink overflow, frames dropped";
      case NFCGPU_EHIP: return "HIP runtime error";
      case NFCGPU_EFULL: return "no free stream slot";
      case NFCGPU_EIO: return "a file could not be opened or written";
      default: return "unknown error";
   }
}

const char *nfcgpu_last_error(nfcgpu_ctx *ctx)
{
   return ctx ? ctx->lastError.c_str() : "";
}

#ifdef NFCGPU_EMULATED_TEST_BUILD
/* TEST HOOKS, only in the emulated build of tests/hostsim (device memory is host memory there): place a stream's sample
 * clock (device record and host mirror) anywhere, e.g. next to the 32-bit wrap, and read it back. The reference offers no
 * way to preset its clock, so the product has none either. */
int nfcgpu_test_set_clock(nfcgpu_ctx *ctx, uint32_t id, uint32_t clock)
{
   if (!ctx || id >= ctx->maxStreams || !ctx->streams[id].open || !ctx->streams[id].initialized)
      return NFCGPU_ESTREAM;
   ctx->dStates[id].clock = clock;
   ctx->streams[id].clock = clock;
   return NFCGPU_OK;
}

int nfcgpu_test_get_clock(nfcgpu_ctx *ctx, uint32_t id, uint32_t *device, uint32_t *mirror)
{
   if (!ctx || id >= ctx->maxStreams || !device || !mirror)
      return NFCGPU_ESTREAM;
   *device = ctx->dStates[id].clock;
   *mirror = ctx->streams[id].clock;
   return NFCGPU_OK;
}
#endif

const char *nfcgpu_version(void)
{
#ifdef NFCGPU_EMULATED_TEST_BUILD
   return "nfcgpu 0.1 (test build of the host runtime on an emulated HIP: not a product library)";
#else
   return "nfcgpu 0.1 (gfx950)";
#endif
}

}

namespace {

/* rows first .. first + count - 1 of a uniform submission resident on the device: contiguous runs of one configuration ->
 * one launch each (normally exactly one) */
int run_rows(nfcgpu_ctx *ctx, uint32_t first, uint32_t count, const uint8_t *devBase, uint64_t devPitch, uint32_t n, uint32_t stride)
{
   int rc = NFCGPU_OK;
   uint32_t i = first;

   while (i < first + count)
   {
      const uint32_t c = ctx->streams[i].config;
      uint32_t j = i;

      while (j < first + count && ctx->streams[j].config == c)
         j++;

      {
         std::vector<WindowedItem> items(j - i);
         for (uint32_t k = i; k < j; k++)
            items[k - i] = WindowedItem {k, devBase + (uint64_t)(k - first) * devPitch, n};

         if (windowed_eligible(ctx, c, items))
         {
            rc = run_windowed(ctx, c, items, stride);
            if (rc)
               return rc;

            i = j;
            continue;
         }
      }

      NfcLaunch L = base_launch(ctx);
      L.works = nullptr;
      L.uniformBase = devBase + (uint64_t)(i - first) * devPitch;
      L.uniformPitch = devPitch;
      L.uniformCount = n;
      L.uniformStride = stride;
      L.firstSlot = i;
      L.slotCount = j - i;

      bool exactPossible = false, exactOnly = true;
      for (uint32_t k = i; k < j; k++)
      {
         const bool exact = exact_zone(ctx->streams[k].clock, n);
         exactPossible = exactPossible || exact;
         exactOnly = exactOnly && exact;
      }

      rc = launch_demod(ctx, c, L, (uint64_t)n * (j - i), exactPossible, exactOnly);
      if (rc)
         return rc;

      for (uint32_t k = i; k < j; k++)
         commit_clock(ctx->streams[k], n);

      i = j;
   }

   return NFCGPU_OK;
}

}
