/*
 * TEST INFRASTRUCTURE — task-level drop-in check (SURVEY.md 8(b) "outer contract", BASELINE config 1).
 *
 * Drives the reference's lab::RadioDecoderTask (compiled from /root/reference where it lies, unmodified) exactly as
 * the Qt app / nfc-rx do: the worker is submitted to an rt::Executor, configured and started through the
 * "radio.decoder.command" subject, fed hw::SignalBuffer objects on "radio.signal.raw" and observed on
 * "radio.decoder.frame". Linked once against the reference decoder (oracle/_ref/task-ref, the CPU plumbing run of
 * BASELINE configs[0]) and once against the GPU shim + libnfcgpu.so (oracle/_ref/task-gpu): same binary otherwise.
 *
 * usage: task-harness [--iq] file.wav [file.wav ...]
 *    --iq   publish SIGNAL_TYPE_RADIO_IQ buffers (interleaved I/Q, |IQ| equal to the capture's magnitude) instead of
 *           magnitude buffers: what a receiver task would hand over if it skipped its host-side magnitude pass
 *           (SURVEY 8(f) rank 2). The GPU decoder demodulates them from IQ; the reference decoder takes no sample from them and never returns
 *           (NfcTech.cpp:30, NfcDecoder.cpp:441), so this mode is for the GPU build only.
 * prints one line per NFC poll/listen frame:
 *    FRAME <file> tech type flags phase rate sampleStart sampleEnd sampleRate hexdata
 * and "DONE <file> <frames> <seconds>" per file.
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <list>
#include <mutex>
#include <string>
#include <thread>

#include <nlohmann/json.hpp>

#include <rt/Event.h>
#include <rt/Executor.h>
#include <rt/Logger.h>
#include <rt/Subject.h>

#include <hw/RecordDevice.h>
#include <hw/SignalBuffer.h>
#include <hw/SignalType.h>

#include <lab/data/RawFrame.h>
#include <lab/tasks/RadioDecoderTask.h>

using json = nlohmann::json;

namespace {

std::mutex frameLock;
std::list<lab::RawFrame> frames;
std::atomic<bool> finished {false};

bool command(rt::Subject<rt::Event> *subject, int code, const json &data)
{
   std::atomic<int> outcome {0};

   subject->next({code, [&outcome] { outcome = 1; }, [&outcome](int, const std::string &) { outcome = -1; }, {{"data", data.dump()}}});

   for (int i = 0; i < 2000 && outcome == 0; i++)
      std::this_thread::sleep_for(std::chrono::milliseconds(5));

   return outcome == 1;
}

int decodeFile(const std::string &path, rt::Subject<rt::Event> *commands, rt::Subject<hw::SignalBuffer> *signal, bool iq)
{
   hw::RecordDevice source(path);

   if (!source.open(hw::RecordDevice::Mode::Read))
   {
      std::fprintf(stderr, "cannot open %s\n", path.c_str());
      return -1;
   }

   const unsigned int channels = std::get<unsigned int>(source.get(hw::SignalDevice::PARAM_CHANNEL_COUNT));
   const unsigned int sampleRate = std::get<unsigned int>(source.get(hw::SignalDevice::PARAM_SAMPLE_RATE));

   {
      std::lock_guard<std::mutex> lock(frameLock);
      frames.clear();
      finished = false;
   }

   const json config = {
      {"enabled", true},
      {"sampleRate", sampleRate},
      {"streamTime", 0},
      {"protocol", {{"nfca", {{"enabled", true}}}, {"nfcb", {{"enabled", true}}}, {"nfcf", {{"enabled", true}}}, {"nfcv", {{"enabled", true}}}}}};

   if (!command(commands, lab::RadioDecoderTask::Configure, config))
      return -2;

   if (!command(commands, lab::RadioDecoderTask::Start, json::object()))
      return -3;

   const auto begin = std::chrono::steady_clock::now();
   unsigned long published = 0;

   while (!source.isEof())
   {
      hw::SignalBuffer samples(65536 * channels, channels, 1, sampleRate, 0, 0, hw::SignalType::SIGNAL_TYPE_RADIO_SAMPLES, 0);

      if (source.read(samples) > 0)
      {
         if (iq && channels == 1)
         {
            /* the sample goes on +I, +Q, -I, -Q in turn (axis changes every 4096 samples): sqrtf(m*m + 0) == |m| */
            const unsigned int count = samples.remaining();
            hw::SignalBuffer complex(count * 2, 2, 1, sampleRate, 0, 0, hw::SignalType::SIGNAL_TYPE_RADIO_IQ, 0);

            for (unsigned int i = 0; i < count; i++)
            {
               const float m = samples[i];
               const unsigned int axis = ((published + i) >> 12) & 3;
               complex.put(axis == 0 ? m : (axis == 2 ? -m : 0.0f)).put(axis == 1 ? m : (axis == 3 ? -m : 0.0f));
            }

            complex.flip();
            published += count;
            signal->next(complex);
         }
         else
         {
            signal->next(samples);
         }
      }

      /* optional pacing (milliseconds per buffer), e.g. 6.5 = the real-time rate of a 10 MS/s receiver */
      if (const char *pace = std::getenv("TASK_HARNESS_PACE_MS"))
         std::this_thread::sleep_for(std::chrono::microseconds((long)(std::atof(pace) * 1000)));
   }

   /* end of stream: the task answers with an empty frame once everything queued before it has been decoded */
   signal->next(hw::SignalBuffer());

   for (int i = 0; i < 60000 && !finished; i++)
      std::this_thread::sleep_for(std::chrono::milliseconds(5));

   const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();

   if (!finished)
      return -4;

   std::lock_guard<std::mutex> lock(frameLock);

   int count = 0;

   for (const auto &frame: frames)
   {
      if (frame.frameType() != lab::FrameType::NfcPollFrame && frame.frameType() != lab::FrameType::NfcListenFrame)
         continue;

      std::string hex;
      char digits[4];

      for (unsigned int i = 0; i < frame.limit(); i++)
      {
         std::snprintf(digits, sizeof(digits), "%02X", (unsigned int)frame.data()[i]);
         hex += digits;
      }

      std::printf("FRAME %s %u %u %u %u %u %lu %lu %lu %s\n", path.c_str(), frame.techType(), frame.frameType(), frame.frameFlags(),
                  frame.framePhase(), frame.frameRate(), frame.sampleStart(), frame.sampleEnd(), frame.sampleRate(), hex.empty() ? "-" : hex.c_str());
      count++;
   }

   std::printf("DONE %s %d %.3f\n", path.c_str(), count, seconds);
   std::fflush(stdout);

   return count;
}

}

int main(int argc, char *argv[])
{
   rt::Logger::init(std::cerr);
   rt::Logger::setRootLevel(rt::Logger::WARN_LEVEL);

   rt::Executor executor(16, 4);

   executor.submit(lab::RadioDecoderTask::construct());

   auto *commands = rt::Subject<rt::Event>::name("radio.decoder.command");
   auto *signal = rt::Subject<hw::SignalBuffer>::name("radio.signal.raw");
   auto *decoded = rt::Subject<lab::RawFrame>::name("radio.decoder.frame");

   auto subscription = decoded->subscribe([](const lab::RawFrame &frame) {
      if (!frame.isValid())
      {
         finished = true;
         return;
      }

      std::lock_guard<std::mutex> lock(frameLock);
      frames.push_back(frame);
   });

   /* let the worker reach its loop */
   std::this_thread::sleep_for(std::chrono::milliseconds(100));

   int status = 0;
   bool iq = false;

   for (int i = 1; i < argc; i++)
   {
      if (std::string(argv[i]) == "--iq")
      {
         iq = true;
         continue;
      }

      if (decodeFile(argv[i], commands, signal, iq) < 0)
      {
         std::fprintf(stderr, "FAILED %s\n", argv[i]);
         status = 1;
      }
   }

   executor.shutdown();

   /* frames hold buffers that return to the reference's static pools (rt::Heap) when released: release them while the
    * pools still exist, not during static destruction */
   {
      std::lock_guard<std::mutex> lock(frameLock);
      frames.clear();
   }

   return status;
}
