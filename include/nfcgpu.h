/*
 * nfcgpu.h — C ABI of libnfcgpu.so, the MI355X (gfx950) implementation of nfc-laboratory's radio
 * demodulation hot path. This is the drop-in boundary: everything above it (lab::NfcDecoder shim,
 * RadioDecoderTask, the Qt app, nfc-rx, test-sdr) is host C++ from the reference, unchanged.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference/src/nfc-lib):
 *
 *   nfcgpu_init / nfcgpu_shutdown      process-wide decoder resources; the reference has none
 *                                      (lab::NfcDecoder::NfcDecoder(), lib-lab/lab-radio/src/main/cpp/NfcDecoder.cpp:75-77,288-290)
 *   nfcgpu_stream_open                 one `lab::NfcDecoder` instance == one stream
 *                                      (lib-lab/lab-radio/src/main/include/lab/nfc/NfcDecoder.h:33-122)
 *   nfcgpu_stream_configure            NfcDecoder::setEnableNfcA/B/F/V, setPowerLevelThreshold,
 *                                      setModulationThresholdNfcX, setCorrelationThresholdNfcX, setSampleRate,
 *                                      setStreamTime (NfcDecoder.cpp:96-286), as driven by
 *                                      RadioDecoderTask::configDecoder (lib-lab/lab-tasks/src/main/cpp/tasks/RadioDecoderTask.cpp:207-366)
 *   nfcgpu_stream_reset                NfcDecoder::initialize() (NfcDecoder.cpp:295-360)
 *   nfcgpu_submit                      NfcDecoder::nextFrames(hw::SignalBuffer) for a valid buffer
 *                                      (NfcDecoder.cpp:374-447); stride 1 = SIGNAL_TYPE_RADIO_SAMPLES magnitude,
 *                                      stride 2 = SIGNAL_TYPE_RADIO_IQ with the IQ->magnitude step of
 *                                      RadioDeviceTask::processQueue (lab-tasks/.../RadioDeviceTask.cpp:547-656) fused in
 *   nfcgpu_submit_batch / _uniform     the same call for many independent streams at once (the reference would run one
 *                                      NfcDecoder per stream on one thread each); inputs may already be resident in HBM
 *   nfcgpu_flush                       NfcDecoder::nextFrames(invalid buffer) -> one carrier-state frame (NfcDecoder.cpp:449-463)
 *   nfcgpu_poll                        the returned std::list<lab::RawFrame> (lab-data/src/main/cpp/RawFrame.cpp:26-98)
 *   nfcgpu_stream_close                ~NfcDecoder
 *
 * Conventions: every function returns 0 on success or a negative NFCGPU_E* code; no exceptions cross the ABI;
 * outputs are caller-allocated; input buffers are never retained past the call that received them unless they are
 * device-resident (then they must stay valid until nfcgpu_sync / nfcgpu_poll returns). One thread per context.
 * There is no CPU fallback: without a usable gfx950 device nfcgpu_init fails with NFCGPU_ENODEV.
 */
#ifndef NFCGPU_H
#define NFCGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFCGPU_OK 0
#define NFCGPU_EINVAL (-1)    /* bad argument */
#define NFCGPU_ENODEV (-2)    /* no usable HIP device / kernel image */
#define NFCGPU_ENOMEM (-3)    /* allocation failed, or more than 256 distinct decoder configurations in use at once */
#define NFCGPU_ESTREAM (-4)   /* unknown or closed stream id */
#define NFCGPU_ERATE (-5)     /* sample rate not decodable (history depth) */
#define NFCGPU_EOVERFLOW (-6) /* frame sink overflowed, frames were dropped */
#define NFCGPU_EHIP (-7)      /* HIP runtime error, see nfcgpu_last_error */
#define NFCGPU_EFULL (-8)     /* no free stream slot */
#define NFCGPU_EIO (-9)       /* a file could not be opened or written (nfcgpu_trace_write*) */

#define NFCGPU_TECH_A 0x1u
#define NFCGPU_TECH_B 0x2u
#define NFCGPU_TECH_F 0x4u
#define NFCGPU_TECH_V 0x8u

#define NFCGPU_LOC_HOST 0u
#define NFCGPU_LOC_DEVICE 1u

typedef struct nfcgpu_ctx nfcgpu_ctx;

/* decoder configuration of one stream; defaults (nfcgpu_default_params) are the reference's
 * (NfcTech.h:347, NfcA.cpp:94-100, NfcB.cpp:103-109, NfcF.cpp:88-94, NfcV.cpp:101-107) */
typedef struct nfcgpu_params
{
   uint32_t sample_rate;          /* Hz; 0 = take it from the first submitted buffer */
   uint32_t tech_mask;            /* NFCGPU_TECH_* */
   int64_t stream_time;           /* reference time added to frame dateTime (NfcDecoder::setStreamTime) */
   float power_level_threshold;
   float corr_threshold[4];       /* A B F V */
   float min_modulation_depth[4]; /* A B F V */
   float max_modulation_depth[4]; /* A B F V */
} nfcgpu_params;

/* one decoded frame == the compared fields of lab::RawFrame plus payload (RawFrame.cpp:82-98) */
typedef struct nfcgpu_frame
{
   uint32_t stream_id;
   uint32_t tech_type;   /* lab::FrameTech  */
   uint32_t frame_type;  /* lab::FrameType  */
   uint32_t frame_flags; /* lab::FrameFlags */
   uint32_t frame_phase; /* lab::FramePhase */
   uint32_t frame_rate;
   uint32_t length;
   uint32_t reserved;
   uint64_t sample_start;
   uint64_t sample_end;
   uint64_t sample_rate;
   uint8_t data[512];
} nfcgpu_frame;

typedef struct nfcgpu_options
{
   uint32_t max_streams;     /* stream slots reserved in HBM (rounded up to 64); default 1024 */
   uint32_t reserved;
   uint64_t frame_sink_bytes; /* device frame sink per sync interval; default 64 MiB */
} nfcgpu_options;

/* many streams, one call. data[i] points to n_samples[i]*stride floats of stream stream_ids[i] */
typedef struct nfcgpu_batch
{
   uint32_t n_streams;
   uint32_t stride;      /* 1 magnitude, 2 interleaved IQ */
   uint32_t location;    /* NFCGPU_LOC_* of every data[i] */
   uint32_t sample_rate; /* Hz */
   const uint32_t *stream_ids;
   const void *const *data;
   const uint32_t *n_samples;
} nfcgpu_batch;

/* accumulated since nfcgpu_stats_reset; kernel_ms is HIP-event time of the demodulation kernel on the context's stream */
typedef struct nfcgpu_stats
{
   uint64_t launches;
   uint64_t samples;
   uint64_t frames;
   uint64_t dropped_frames;
   double kernel_ms;
   /* time-parallel path: scan kernel time and samples (profiling on), windowed decode kernel time,
    * windows decoded (lanes, repeats included), decode passes, streams of submissions that took it / fell back */
   double scan_ms;
   double window_ms;
   uint64_t scan_samples;
   uint64_t windows;
   uint64_t window_passes;
   uint64_t windowed_streams;
   uint64_t fallback_streams;
   uint64_t scan_repairs; /* scan chunks walked a second time because their warm-up had not reached the true state */
   /* round 3 (read through nfcgpu_stats_get_sized by callers built against this header; nfcgpu_stats_get fills the
    * fields above only): the wave decoder's kernel - HIP-event time of all its launches and their number - and the walk
    * that writes the front-end planes for it */
   double wave_ms;
   uint64_t wave_launches;
   double planes_ms;
   /* round 4: the time the wave decoder's kernel was running at all since nfcgpu_stats_reset - the union of its launch
    * intervals (the carry lanes of a pass run on a second HIP stream beside the speculative lanes, so wave_ms, the sum of
    * the launch durations, counts the overlap twice). Profiling on; measured from the reset on. */
   double wave_busy_ms;
} nfcgpu_stats;

#define NFCGPU_STATS_SIZE_V2 104u /* bytes of nfcgpu_stats up to and including scan_repairs: what nfcgpu_stats_get writes */

void nfcgpu_default_params(nfcgpu_params *params);

int nfcgpu_init(int device, const nfcgpu_options *options, nfcgpu_ctx **ctx);
int nfcgpu_shutdown(nfcgpu_ctx *ctx);

int nfcgpu_stream_open(nfcgpu_ctx *ctx, const nfcgpu_params *params, uint32_t *stream_id);
int nfcgpu_stream_open_many(nfcgpu_ctx *ctx, const nfcgpu_params *params, uint32_t count, uint32_t *first_stream_id);
/* The setters of lab::NfcDecoder (NfcDecoder.cpp:96-286). The enable mask and the per-technology thresholds take effect
 * with the next sample; the power level moves the detectors' gate at once and the carrier thresholds at the next
 * initialisation (NfcDecoder.cpp:327-329); sample_rate is only stored (0 leaves it as it is), exactly like
 * setSampleRate(): nothing is derived from it until the stream (re)initialises. */
int nfcgpu_stream_configure(nfcgpu_ctx *ctx, uint32_t stream_id, const nfcgpu_params *params);
/* initialize() (NfcDecoder.cpp:295-360): clock, detectors, protocol state and everything derived from the sample rate and
 * the power level stored at this moment start over; envelope, filter and carrier state carry on. Applied when the next
 * buffer arrives; an nfcgpu_flush() in between already sees the reset clock. */
int nfcgpu_stream_reset(nfcgpu_ctx *ctx, uint32_t stream_id);
int nfcgpu_stream_close(nfcgpu_ctx *ctx, uint32_t stream_id);

/* nextFrames(valid buffer) (NfcDecoder.cpp:374-447) without the frame collection (nfcgpu_poll). A buffer whose sample rate
 * differs from the stored one stores it and re-initialises the stream first, also when it is empty (n_samples 0); a
 * buffer at the stored rate does not, whatever the stored rate was derived-from last (see nfcgpu_stream_configure). */
/* Submissions are asynchronous: the calls return once the work is enqueued on the context's HIP stream. Host memory
 * (NFCGPU_LOC_HOST, nfcgpu_submit) has been copied into a pinned staging buffer by then and is never retained; device
 * memory (NFCGPU_LOC_DEVICE) is read in place and must stay as it is until the next nfcgpu_sync / nfcgpu_poll /
 * nfcgpu_flush / nfcgpu_pending of the context. (Long grid-aligned submissions that take the time-parallel path are
 * complete when the call returns.) */
int nfcgpu_submit(nfcgpu_ctx *ctx, uint32_t stream_id, const float *data, uint32_t n_samples, uint32_t stride, uint32_t sample_rate);
int nfcgpu_submit_batch(nfcgpu_ctx *ctx, const nfcgpu_batch *batch);
/* streams first..first+count-1; stream i reads n_samples*stride floats at base + i*pitch_bytes */
int nfcgpu_submit_uniform(nfcgpu_ctx *ctx, uint32_t first_stream_id, uint32_t count, const void *base, uint64_t pitch_bytes,
                          uint32_t n_samples, uint32_t stride, uint32_t location, uint32_t sample_rate);

/* Magnitude of interleaved float IQ, out[i] = sqrtf(I*I + Q*Q) with the reference's roundings (products and sum
 * rounded separately, correctly rounded root): the conversion RadioDeviceTask applies before publishing a
 * SIGNAL_TYPE_RADIO_SAMPLES buffer (RadioDeviceTask.cpp:547-656, scalar form 626-642). The decoder entry points do
 * this on the fly for stride-2 input; this call exposes the same device function for hosts that also want the
 * magnitudes (storage, display) and for bit-exact testing. `location` applies to both pointers; the call returns
 * when `out` is complete. */
int nfcgpu_magnitude(nfcgpu_ctx *ctx, const float *iq, uint64_t n_samples, float *out, uint32_t location);

/* Adaptive resampling of magnitude buffers for display, the radio branch of the reference's SignalResamplingTask
 * (SignalResamplingTask.cpp:168-226: `processRadioSignal`, the other per-sample consumer of "radio.signal.raw"): every
 * buffer of n_samples floats is reduced to (value, sample offset) control points wherever the sample departs from its
 * 51-sample centred mean by more than 0.005, or every 255 samples; each buffer is independent (the running mean
 * restarts with it), the output of buffer b is written as float pairs at out + b * out_pitch_bytes and its pair count
 * to counts[b] — the contents of the SIGNAL_TYPE_RADIO_SIGNAL buffer the reference publishes on "adaptive.signal".
 * capacity_pairs bounds what is written per buffer (the worst case is n_samples + n_samples / 255 + 2 pairs); a buffer
 * that needs more keeps counting and the call returns NFCGPU_EOVERFLOW. n_samples >= 25. `out` and out_pitch_bytes are 8-byte aligned. `location` applies to `in`,
 * `out` and `counts` alike. */
int nfcgpu_resample_radio(nfcgpu_ctx *ctx, const float *in, uint64_t in_pitch_bytes, uint32_t n_buffers, uint32_t n_samples,
                          float *out, uint64_t out_pitch_bytes, uint32_t capacity_pairs, uint32_t *counts, uint32_t location);

/* nextFrames(invalid buffer) (NfcDecoder.cpp:449-463): queues one carrier-state frame stamped with the stream's clock */
int nfcgpu_flush(nfcgpu_ctx *ctx, uint32_t stream_id);
/* waits for everything submitted, then moves the frames of the frame sink to the per-stream queues. poll / pending /
 * flush call it; with nothing in flight and nothing to collect none of them touches the device. */
int nfcgpu_sync(nfcgpu_ctx *ctx);
/* Frames wait in a queue per stream until they are polled: nfcgpu_sync moves the frames of every stream there, so a stream
 * that is never polled keeps what it has produced (memory grows with its frames; nfcgpu_stream_close releases it). */
int nfcgpu_poll(nfcgpu_ctx *ctx, uint32_t stream_id, nfcgpu_frame *out, uint32_t capacity, uint32_t *count);
int nfcgpu_pending(nfcgpu_ctx *ctx, uint32_t stream_id, uint32_t *count);

/* Trace files: decoded frames in the format the reference application opens ("open trace") and
 * tools/py_nfclab reads - a gzip-compressed tar archive with one entry frame.json, as TraceStorageTask writes it
 * (lab-tasks/src/main/cpp/tasks/TraceStorageTask.cpp:461-520; read back by :380-449). range_start / range_end in seconds of
 * stream time keep the frames inside the range and shift their times and sample numbers to its start (:461-483, the
 * range of the "write file" command :211-240); 0, 0 = every frame. *written (may be NULL) = frames in the file.
 *   nfcgpu_trace_write_frames  any frames the caller holds (no context, no device);
 *   nfcgpu_trace_write         the frames the device has decoded for a stream and that wait in its queue (nfcgpu_poll
 *                              order; implies nfcgpu_sync); they stay queued. dateTime = the stream's stream_time + timeStart.
 *                              NFCGPU_EINVAL while the sink is held (nfcgpu_sink_hold: nothing is drained into the queues then,
 *                              the file would come out empty).
 * One departure from the reference's writer: with a range it keeps the frames that lie inside it with both ends, as the
 * reference does (timeEnd <= range end); 0, 0 means every frame, where the reference's command always carries a range.
 * "length" is the number of bytes in frameData (at most the 512 a frame record holds). A file that cannot be opened or
 * written is NFCGPU_EIO. */
int nfcgpu_trace_write_frames(const char *path, const nfcgpu_frame *frames, uint32_t count, int64_t stream_time, double range_start, double range_end,
                              uint32_t *written);
int nfcgpu_trace_write(nfcgpu_ctx *ctx, uint32_t stream_id, const char *path, double range_start, double range_end, uint32_t *written);

/* device-resident view of the frames produced since the last sync: packed records of 32-bit words
 * [stream_id, tech, type, flags, phase, rate, start, end, length, payload...]; used for RCCL frame gathers */
int nfcgpu_sink_device_view(nfcgpu_ctx *ctx, const void **words, const void **cursor_words, uint64_t *capacity_words);
/* use caller-owned device memory as the frame sink (e.g. a torch tensor that RCCL can gather directly):
 * `words` holds capacity_words 32-bit words, `ctl` holds 4 words ([0] cursor in words, [1] dropped frames).
 * Pass words == NULL to return to the context's own sink. Implies a sync + rewind. */
int nfcgpu_sink_attach(nfcgpu_ctx *ctx, void *words, uint64_t capacity_words, void *ctl);
/* when enabled, nfcgpu_sync leaves the sink untouched (no host drain) until nfcgpu_sink_rewind: the caller reads the
 * packed records itself. nfcgpu_poll / nfcgpu_flush then only deal with the per-stream queues filled before; a held sink
 * is meant for hosts that collect frames in bulk (bench.py, RCCL gathers), not to be mixed with polling. */
int nfcgpu_sink_hold(nfcgpu_ctx *ctx, int hold);
int nfcgpu_sink_rewind(nfcgpu_ctx *ctx);

/* ---- multi-GPU: one context (one process) per GPU, streams sharded over the ranks; the only exchange is the gather of
 * the decoded frames, done here in C++ over RCCL so that a host that is not Python (the Qt application,
 * nfc-rx) has it too. The unique id is made on rank 0 and carried to the other ranks by whatever the host uses to start
 * its processes (MPI, sockets, a file, torch.distributed). librccl.so is looked up when the first of these is called. */
#define NFCGPU_UNIQUE_ID_BYTES 128
int nfcgpu_comm_unique_id(void *id128);
int nfcgpu_comm_init(nfcgpu_ctx *ctx, const void *id128, int rank, int n_ranks);
int nfcgpu_comm_destroy(nfcgpu_ctx *ctx);
/* All-gather of every rank's packed frame records (the context's frame sink as it stands: nfcgpu_sink_hold must be on,
 * so that nothing has been drained - NFCGPU_EINVAL otherwise): one ncclAllGather of (word count, receive capacity) per
 * rank, on which every rank takes the same go / no-go decision (NFCGPU_ENOMEM everywhere when some rank's buffer is too
 * small), then the records at their exact sizes (one ncclBroadcast per rank, grouped). `gathered` is a device buffer of
 * capacity_words words; counts_host[r] = words of rank r.
 *   nfcgpu_gather_frames_packed  rank r's records start at the sum of counts_host[0 .. r-1] (needs the sum of the counts);
 *   nfcgpu_gather_frames         the layout of rounds 1-2: rank r's records start at r * *stride_words, *stride_words = the
 *                                largest count (needs n_ranks times that). Round 3 returned the packed layout with a stride
 *                                of 0 under this symbol; a caller built against the older header reads every rank at
 *                                r * stride, so the padded layout is what this symbol keeps.
 * Record format of the sink: [stream, tech, type, flags, phase, rate, start, end, length, payload words]. */
int nfcgpu_gather_frames(nfcgpu_ctx *ctx, void *gathered, uint64_t capacity_words, uint32_t *counts_host, uint64_t *stride_words);
int nfcgpu_gather_frames_packed(nfcgpu_ctx *ctx, void *gathered, uint64_t capacity_words, uint32_t *counts_host);

/* streaming-read bandwidth of this GPU over `bytes` of device memory (16-byte loads per lane, grid sized to the chip):
 * the measured denominator of the HBM roofline, next to the vendor peak. Best of `repeats` passes, GB/s. */
int nfcgpu_read_bandwidth(nfcgpu_ctx *ctx, const void *device_ptr, uint64_t bytes, uint32_t repeats, double *gbps);

int nfcgpu_stats_get(nfcgpu_ctx *ctx, nfcgpu_stats *stats); /* writes NFCGPU_STATS_SIZE_V2 bytes (the struct of rounds 1-2) */
/* writes min(size, sizeof(nfcgpu_stats)) bytes: pass sizeof(nfcgpu_stats) of the header the caller was built with */
int nfcgpu_stats_get_sized(nfcgpu_ctx *ctx, void *stats, uint32_t size);
int nfcgpu_stats_reset(nfcgpu_ctx *ctx);
int nfcgpu_profile(nfcgpu_ctx *ctx, int enable);

void *nfcgpu_hip_stream(nfcgpu_ctx *ctx);
const char *nfcgpu_strerror(int code);
const char *nfcgpu_last_error(nfcgpu_ctx *ctx);
const char *nfcgpu_version(void);

#ifdef __cplusplus
}
#endif

#endif
