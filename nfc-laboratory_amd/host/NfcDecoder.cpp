/*
 * Drop-in implementation of lab::NfcDecoder on top of libnfcgpu.so.
 *
 * Compiled against the reference's own public headers
 *   src/nfc-lib/lib-lab/lab-radio/src/main/include/lab/nfc/NfcDecoder.h  (class declaration, unchanged)
 *   src/nfc-lib/lib-hw/hw-dev/src/main/include/hw/SignalBuffer.h
 *   src/nfc-lib/lib-lab/lab-data/src/main/include/lab/data/RawFrame.h
 * it provides exactly the symbols that RadioDecoderTask.o, nfc-rx and test-sdr import from lab-radio
 * (SURVEY.md 8(b)), so those link and run unchanged; the per-sample work happens in HIP kernels behind the
 * C ABI of include/nfcgpu.h. One NfcDecoder instance == one nfcgpu stream; all instances of a process share
 * one nfcgpu context (one GPU), serialised by one lock (a decoder is still used by one thread at a time, like the
 * reference's, but different decoders may live on different threads). There is no CPU decoding path: if the GPU
 * runtime cannot be initialised the constructor throws, and a buffer the GPU side refuses (sample rate beyond the
 * history depth, HIP error) makes nextFrames() throw instead of silently dropping samples.
 *
 * This file replaces src/nfc-lib/lib-lab/lab-radio/src/main/cpp/NfcDecoder.cpp (+ NfcTech.cpp, tech/*.cpp)
 * in the reference's lab-radio library; see INTEGRATION.md.
 */
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include <hw/SignalType.h>
#include <hw/SignalBuffer.h>

#include <lab/data/RawFrame.h>
#include <lab/nfc/NfcDecoder.h>

#include <nfcgpu.h>

namespace lab {

namespace {

struct SharedContext
{
   nfcgpu_ctx *ctx = nullptr;
   bool closed = false;
   std::mutex mutex;
   std::recursive_mutex use; /* the context is single-threaded: every call into it holds this */

   /* Decoders owned by long-lived objects (the RadioDecoderTask worker, static subjects) may be destroyed while the
    * process is exiting, after the GPU runtime has begun to unload. The context is therefore released by an atexit
    * hook registered after the runtime's own (it runs before it); decoders destroyed later find it closed. */
   static void atExit();

   bool alive()
   {
      std::lock_guard<std::mutex> lock(mutex);
      return ctx != nullptr && !closed;
   }

   nfcgpu_ctx *get()
   {
      std::lock_guard<std::mutex> lock(mutex);

      if (closed)
         throw std::runtime_error("nfcgpu context already shut down");

      if (!ctx)
      {
         nfcgpu_options options {};
         const char *device = std::getenv("NFCGPU_DEVICE");
         const char *streams = std::getenv("NFCGPU_MAX_STREAMS");
         options.max_streams = streams ? (uint32_t)std::atoi(streams) : 64;

         int rc = nfcgpu_init(device ? std::atoi(device) : 0, &options, &ctx);

         if (rc != NFCGPU_OK)
            throw std::runtime_error(std::string("nfcgpu_init failed: ") + nfcgpu_strerror(rc));

         std::atexit(&SharedContext::atExit);
      }

      return ctx;
   }
};

SharedContext shared;

void SharedContext::atExit()
{
   std::lock_guard<std::recursive_mutex> use(shared.use);
   std::lock_guard<std::mutex> lock(shared.mutex);

   if (shared.ctx)
      nfcgpu_shutdown(shared.ctx);

   shared.ctx = nullptr;
   shared.closed = true;
}

}

struct NfcDecoder::Impl
{
   nfcgpu_ctx *ctx;
   uint32_t stream = 0;
   nfcgpu_params params {};
   bool debugEnabled = false;
   bool dirty = false; /* params changed since last push to the device side */

   /* Block mode (NFCGPU_SHIM_BLOCK=<samples>, off by default). One stream fills one lane of one wavefront, so a 65536-sample
    * buffer at a time runs at a fraction of real time; the library's time-parallel path needs long submissions. In block
    * mode the buffers of nextFrames() are collected until `blockSamples` are there, submitted in one piece (asynchronous),
    * and the frames of a block are handed out when the next block is submitted (or at the end of the stream): the GPU
    * decodes block k while the caller produces block k+1. Frames, their order and their sample clocks are unchanged; they
    * are delivered up to two blocks later than the reference would deliver them. Anything that takes effect "from the
    * next buffer on" (parameters, initialize(), a new sample rate, a buffer of another kind) submits what is pending first.
    * The reference's interface can only hand frames out as the return value of nextFrames(), and RadioDecoderTask ends a
    * stream with cleanup(), not with nextFrames({}) (RadioDecoderTask.cpp:381-397): what is still pending then has nowhere
    * to go. A buffer shorter than the one before it - the last buffer of a file - is therefore taken as the end of the
    * stream: everything is submitted and collected before the call returns. A host that ends a stream on a full buffer
    * has to call nextFrames({}) before cleanup(). */
   size_t blockSamples = 0;
   bool blockAuto = false; /* NFCGPU_SHIM_BLOCK=auto: the block grows (from 2^17 samples, doubling, up to 2^22) whenever decoding a block took
                              longer than 0.8 of the time its samples span at their own rate - the decoder is not keeping up with the
                              receiver, and longer submissions are what the time-parallel path is faster on; it never shrinks */
   std::vector<float> pending;
   unsigned int pendingStride = 1;
   unsigned int pendingRate = 0;
   unsigned int lastCount = 0; /* samples of the buffer before this one */
   std::chrono::steady_clock::time_point pendingSince {}; /* when the first samples of the pending block arrived */
   double blockMillis = 250.0; /* NFCGPU_SHIM_BLOCK_MS */
   std::list<RawFrame> backlog; /* frames collected at a moment that was not a nextFrames() call */
   long submittedStreamTime = 0; /* streamTime() in force for the frames not collected yet */

   Impl() : ctx(shared.get())
   {
      std::lock_guard<std::recursive_mutex> use(shared.use);

      nfcgpu_default_params(&params);

      int rc = nfcgpu_stream_open(ctx, &params, &stream);

      if (rc != NFCGPU_OK)
         throw std::runtime_error(std::string("nfcgpu_stream_open failed: ") + nfcgpu_strerror(rc));

      if (const char *block = std::getenv("NFCGPU_SHIM_BLOCK"))
      {
         if (std::string(block) == "auto")
         {
            blockAuto = true;
            blockSamples = (size_t)1 << 17;
         }
         else
            blockSamples = (size_t)std::strtoull(block, nullptr, 10);
      }
      if (const char *ms = std::getenv("NFCGPU_SHIM_BLOCK_MS"))
         blockMillis = std::strtod(ms, nullptr);
   }

   ~Impl()
   {
      std::lock_guard<std::recursive_mutex> use(shared.use);

      if (shared.alive())
         nfcgpu_stream_close(ctx, stream);
   }

   /* the reference's interface has no error channel: refuse loudly rather than lose samples */
   void check(int rc, const char *what)
   {
      if (rc == NFCGPU_EOVERFLOW)
         std::fprintf(stderr, "nfcgpu: %s: %s\n", what, nfcgpu_last_error(ctx));
      else if (rc != NFCGPU_OK)
         throw std::runtime_error(std::string("nfcgpu: ") + what + ": " + nfcgpu_strerror(rc) + " (" + nfcgpu_last_error(ctx) + ")");
   }

   void push()
   {
      if (dirty)
      {
         dispatch(); /* what has been handed over so far is decoded with the parameters it was handed over under */
         check(nfcgpu_stream_configure(ctx, stream, &params), "stream_configure");
         dirty = false;
      }
   }

   /* block mode: submit what is pending; the frames of everything submitted before go to the backlog */
   void dispatch()
   {
      if (pending.empty())
         return;

      backlog.splice(backlog.end(), collect());

      const size_t count = pending.size() / pendingStride;
      const auto began = std::chrono::steady_clock::now();

      check(nfcgpu_submit(ctx, stream, pending.data(), (uint32_t)count, pendingStride, pendingRate), "submit");
      submittedStreamTime = (long)params.stream_time;
      pending.clear();

      /* (a submission of host samples returns when it has been decoded: its duration is the decode's) */
      if (blockAuto && pendingRate && count >= blockSamples && blockSamples < ((size_t)1 << 22))
      {
         const double took = std::chrono::duration<double>(std::chrono::steady_clock::now() - began).count();

         if (took > 0.8 * (double)count / (double)pendingRate)
            blockSamples *= 2;
      }
   }

   void setTech(uint32_t bit, bool enabled)
   {
      params.tech_mask = enabled ? (params.tech_mask | bit) : (params.tech_mask & ~bit);
      dirty = true;
   }

   void setDepth(int tech, float min, float max)
   {
      if (!std::isnan(min))
         params.min_modulation_depth[tech] = min;
      if (!std::isnan(max))
         params.max_modulation_depth[tech] = max;
      dirty = true;
   }

   void setCorrelation(int tech, float value)
   {
      if (!std::isnan(value))
         params.corr_threshold[tech] = value;
      dirty = true;
   }

   std::list<RawFrame> collect()
   {
      std::list<RawFrame> frames;
      std::vector<nfcgpu_frame> chunk(64);
      uint32_t count = 0;

      do
      {
         check(nfcgpu_poll(ctx, stream, chunk.data(), (uint32_t)chunk.size(), &count), "poll");

         for (uint32_t i = 0; i < count; i++)
         {
            const nfcgpu_frame &f = chunk[i];

            RawFrame frame(f.tech_type, f.frame_type);

            frame.setFrameRate(f.frame_rate);
            frame.setFramePhase(f.frame_phase);
            frame.setFrameFlags(f.frame_flags);
            frame.setSampleStart(f.sample_start);
            frame.setSampleEnd(f.sample_end);
            frame.setSampleRate(f.sample_rate);
            frame.setTimeStart(static_cast<double>(f.sample_start) / static_cast<double>(f.sample_rate));
            frame.setTimeEnd(static_cast<double>(f.sample_end) / static_cast<double>(f.sample_rate));
            frame.setDateTime(submittedStreamTime + frame.timeStart());
            frame.put(f.data, f.length).flip();

            frames.push_back(frame);
         }
      }
      while (count == chunk.size());

      return frames;
   }
};

NfcDecoder::NfcDecoder() : impl(std::make_shared<Impl>())
{
}

void NfcDecoder::initialize()
{
   std::lock_guard<std::recursive_mutex> use(shared.use);

   if (!shared.alive())
      return;

   impl->push();
   impl->dispatch();
   impl->check(nfcgpu_stream_reset(impl->ctx, impl->stream), "stream_reset");
   impl->lastCount = 0; /* (block mode: the next buffer is not a "shorter one" than whatever came before the reset) */
}

/* The reference's cleanup() has nothing to do. In block mode samples may still be waiting for their block to fill when
 * the host stops the decoder (RadioDecoderTask calls cleanup() on the invalid buffer that ends a stream, never
 * nextFrames({}): RadioDecoderTask.cpp:390-399): they are decoded now, and their frames are kept for the next
 * nextFrames() call - the interface has no other way out for them (a host that wants them at once calls
 * nextFrames({}) before cleanup()). */
void NfcDecoder::cleanup()
{
   std::lock_guard<std::recursive_mutex> use(shared.use);

   if (!shared.alive() || !impl->blockSamples)
      return;

   impl->push();
   impl->dispatch();
   impl->backlog.splice(impl->backlog.end(), impl->collect());
   impl->lastCount = 0;
}

std::list<RawFrame> NfcDecoder::nextFrames(hw::SignalBuffer samples)
{
   std::lock_guard<std::recursive_mutex> use(shared.use);

   if (!shared.alive())
      return {};

   impl->push();

   if (samples.isValid())
   {
      /* like the reference, only magnitude buffers advance the decoder (NfcTech.cpp:30); interleaved IQ buffers
       * are accepted as an extension and demodulated from IQ on the GPU */
      const unsigned int type = samples.type();
      const unsigned int stride = type == hw::SignalType::SIGNAL_TYPE_RADIO_IQ ? 2 : 1;

      if (type == hw::SignalType::SIGNAL_TYPE_RADIO_SAMPLES || type == hw::SignalType::SIGNAL_TYPE_RADIO_IQ)
      {
         const unsigned int count = samples.remaining() / stride;

         if (impl->blockSamples && count)
         {
            /* block mode: a buffer of another kind or rate closes the block before it */
            if (!impl->pending.empty() && (impl->pendingStride != stride || impl->pendingRate != samples.sampleRate()))
               impl->dispatch();

            impl->pendingStride = stride;
            impl->pendingRate = samples.sampleRate();
            impl->pending.insert(impl->pending.end(), samples.ptr(), samples.ptr() + (size_t)count * stride);
            impl->params.sample_rate = samples.sampleRate();

            const bool last = count < impl->lastCount;
            impl->lastCount = count;

            /* a block is also closed when its first samples have waited long enough (a slow or bursty receiver must not
             * hold frames back without bound: NFCGPU_SHIM_BLOCK_MS, default 250 ms of wall clock) */
            const auto now = std::chrono::steady_clock::now();
            if (impl->pending.size() == (size_t)count * stride)
               impl->pendingSince = now;
            const bool stale = std::chrono::duration<double, std::milli>(now - impl->pendingSince).count() > impl->blockMillis;

            if (impl->pending.size() / stride >= impl->blockSamples || last || stale)
               impl->dispatch();

            if (last)
               impl->backlog.splice(impl->backlog.end(), impl->collect()); /* waits for the block just submitted */

            std::list<RawFrame> frames;
            frames.swap(impl->backlog);
            return frames;
         }

         impl->dispatch();

         /* an empty buffer still counts: a sample rate that differs from the stored one re-initialises the decoder
          * at this point (NfcDecoder.cpp:383-388), with the thresholds set at this point */
         impl->check(nfcgpu_submit(impl->ctx, impl->stream, samples.ptr(), count, stride, samples.sampleRate()), "submit");
         impl->submittedStreamTime = (long)impl->params.stream_time;

         impl->params.sample_rate = samples.sampleRate();
      }
   }
   else
   {
      impl->dispatch();
      impl->backlog.splice(impl->backlog.end(), impl->collect());
      impl->submittedStreamTime = (long)impl->params.stream_time;
      impl->check(nfcgpu_flush(impl->ctx, impl->stream), "flush");
   }

   std::list<RawFrame> frames;
   frames.swap(impl->backlog);
   frames.splice(frames.end(), impl->collect());
   return frames;
}

bool NfcDecoder::isDebugEnabled() const { return impl->debugEnabled; }
void NfcDecoder::setEnableDebug(bool enabled) { impl->debugEnabled = enabled; /* signal debug taps are a CPU-oracle facility */ }

bool NfcDecoder::isNfcAEnabled() const { return impl->params.tech_mask & NFCGPU_TECH_A; }
void NfcDecoder::setEnableNfcA(bool enabled) { impl->setTech(NFCGPU_TECH_A, enabled); }
bool NfcDecoder::isNfcBEnabled() const { return impl->params.tech_mask & NFCGPU_TECH_B; }
void NfcDecoder::setEnableNfcB(bool enabled) { impl->setTech(NFCGPU_TECH_B, enabled); }
bool NfcDecoder::isNfcFEnabled() const { return impl->params.tech_mask & NFCGPU_TECH_F; }
void NfcDecoder::setEnableNfcF(bool enabled) { impl->setTech(NFCGPU_TECH_F, enabled); }
bool NfcDecoder::isNfcVEnabled() const { return impl->params.tech_mask & NFCGPU_TECH_V; }
void NfcDecoder::setEnableNfcV(bool enabled) { impl->setTech(NFCGPU_TECH_V, enabled); }

long NfcDecoder::sampleRate() const { return impl->params.sample_rate; }

void NfcDecoder::setSampleRate(long sampleRate)
{
   /* the reference only stores the value; the decoder re-initialises when a buffer with another rate arrives */
   impl->params.sample_rate = (uint32_t)sampleRate;
   impl->dirty = true;
}

long NfcDecoder::streamTime() const { return (long)impl->params.stream_time; }
void NfcDecoder::setStreamTime(long referenceTime) { impl->params.stream_time = referenceTime; impl->dirty = true; }

float NfcDecoder::powerLevelThreshold() const { return impl->params.power_level_threshold; }
void NfcDecoder::setPowerLevelThreshold(float value) { impl->params.power_level_threshold = value; impl->dirty = true; }

float NfcDecoder::modulationThresholdNfcAMin() const { return impl->params.min_modulation_depth[0]; }
float NfcDecoder::modulationThresholdNfcAMax() const { return impl->params.max_modulation_depth[0]; }
void NfcDecoder::setModulationThresholdNfcA(float min, float max) { impl->setDepth(0, min, max); }
float NfcDecoder::modulationThresholdNfcBMin() const { return impl->params.min_modulation_depth[1]; }
float NfcDecoder::modulationThresholdNfcBMax() const { return impl->params.max_modulation_depth[1]; }
void NfcDecoder::setModulationThresholdNfcB(float min, float max) { impl->setDepth(1, min, max); }
float NfcDecoder::modulationThresholdNfcFMin() const { return impl->params.min_modulation_depth[2]; }
float NfcDecoder::modulationThresholdNfcFMax() const { return impl->params.max_modulation_depth[2]; }
void NfcDecoder::setModulationThresholdNfcF(float min, float max) { impl->setDepth(2, min, max); }
float NfcDecoder::modulationThresholdNfcVMin() const { return impl->params.min_modulation_depth[3]; }
float NfcDecoder::modulationThresholdNfcVMax() const { return impl->params.max_modulation_depth[3]; }
void NfcDecoder::setModulationThresholdNfcV(float min, float max) { impl->setDepth(3, min, max); }

float NfcDecoder::correlationThresholdNfcA() const { return impl->params.corr_threshold[0]; }
void NfcDecoder::setCorrelationThresholdNfcA(float value) { impl->setCorrelation(0, value); }
float NfcDecoder::correlationThresholdNfcB() const { return impl->params.corr_threshold[1]; }
void NfcDecoder::setCorrelationThresholdNfcB(float value) { impl->setCorrelation(1, value); }
float NfcDecoder::correlationThresholdNfcF() const { return impl->params.corr_threshold[2]; }
void NfcDecoder::setCorrelationThresholdNfcF(float value) { impl->setCorrelation(2, value); }
float NfcDecoder::correlationThresholdNfcV() const { return impl->params.corr_threshold[3]; }
void NfcDecoder::setCorrelationThresholdNfcV(float value) { impl->setCorrelation(3, value); }

}
