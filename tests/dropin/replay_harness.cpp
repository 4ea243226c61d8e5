/*
 * TEST INFRASTRUCTURE - the application's replay pipeline: the reference's SignalStorageTask reads a
 * WAV capture - one channel of magnitudes, or two channels of I/Q which it turns into magnitudes itself
 * (SignalStorageTask.cpp:372-440, the SSE2 twin of RadioDeviceTask's conversion) - and publishes radio.signal.raw; the
 * reference's RadioDecoderTask, in the same executor, decodes that subject. Both tasks are the reference's code, compiled
 * where it lies; the binary is linked once with the reference decoder (oracle/_ref/replay-ref) and once with the
 * lab::NfcDecoder shim on libnfcgpu.so (oracle/_ref/replay-gpu).
 *
 *   replay-harness capture.wav sampleRate [magnitudes.f32]
 * prints  FRAME tech type flags phase rate sampleStart sampleEnd sampleRate hexdata|-   per decoded frame (carrier frames
 * included) and DONE <frames> <samples>; with a third argument the magnitudes the storage task published are written
 * to that file as raw floats (the reference's own IQ -> magnitude results, the yardstick of nfcgpu_magnitude).
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
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

#include <hw/SignalBuffer.h>
#include <hw/SignalType.h>

#include <lab/data/RawFrame.h>
#include <lab/tasks/RadioDecoderTask.h>
#include <lab/tasks/SignalStorageTask.h>

using json = nlohmann::json;

static bool command(rt::Subject<rt::Event> *subject, int code, const json &data)
{
   std::atomic<int> outcome {0};

   subject->next({code, [&outcome] { outcome = 1; }, [&outcome](int, const std::string &) { outcome = -1; }, {{"data", data.dump()}}});

   for (int i = 0; i < 2000 && outcome == 0; i++)
      std::this_thread::sleep_for(std::chrono::milliseconds(5));

   return outcome == 1;
}

int main(int argc, char *argv[])
{
   if (argc < 3)
      return 2;

   rt::Logger::init(std::cerr);
   rt::Logger::setRootLevel(rt::Logger::WARN_LEVEL);

   std::mutex lock;
   std::list<lab::RawFrame> frames;
   std::atomic<bool> finished {false};
   std::atomic<unsigned long> samples {0};
   std::ofstream dump;

   if (argc > 3)
      dump.open(argv[3], std::ios::binary);

   rt::Executor executor(16, 4);
   executor.submit(lab::RadioDecoderTask::construct());
   executor.submit(lab::SignalStorageTask::construct());

   auto *decoderCommands = rt::Subject<rt::Event>::name("radio.decoder.command");
   auto *recorderCommands = rt::Subject<rt::Event>::name("recorder.command");
   auto *raw = rt::Subject<hw::SignalBuffer>::name("radio.signal.raw");
   auto *decoded = rt::Subject<lab::RawFrame>::name("radio.decoder.frame");

   auto frameSubscription = decoded->subscribe([&](const lab::RawFrame &frame) {
      if (!frame.isValid())
      {
         finished = true;
         return;
      }

      std::lock_guard<std::mutex> guard(lock);
      frames.push_back(frame);
   });

   auto rawSubscription = raw->subscribe([&](const hw::SignalBuffer &buffer) {
      if (!buffer.isValid())
         return;

      samples += buffer.elements();

      if (dump.is_open())
         dump.write(reinterpret_cast<const char *>(buffer.data()), buffer.elements() * sizeof(float));
   });

   std::this_thread::sleep_for(std::chrono::milliseconds(100));

   const json config = {
      {"enabled", true},
      {"sampleRate", std::atol(argv[2])},
      {"streamTime", 0},
      {"protocol", {{"nfca", {{"enabled", true}}}, {"nfcb", {{"enabled", true}}}, {"nfcf", {{"enabled", true}}}, {"nfcv", {{"enabled", true}}}}}};

   int status = 0;

   if (!command(decoderCommands, lab::RadioDecoderTask::Configure, config) || !command(decoderCommands, lab::RadioDecoderTask::Start, json::object()))
      status = 3;

   if (!status && !command(recorderCommands, lab::SignalStorageTask::Read, {{"fileName", argv[1]}}))
      status = 4;

   for (int i = 0; !status && i < 120000 && !finished; i++)
      std::this_thread::sleep_for(std::chrono::milliseconds(5));

   if (!status && !finished)
      status = 5;

   {
      std::lock_guard<std::mutex> guard(lock);

      for (const lab::RawFrame &frame: frames)
      {
         std::string hex;
         char digits[4];

         for (unsigned int i = 0; i < frame.limit(); i++)
         {
            std::snprintf(digits, sizeof(digits), "%02x", (unsigned int)frame.data()[i]);
            hex += digits;
         }

         std::printf("FRAME %u %u %u %u %u %lu %lu %lu %s\n", frame.techType(), frame.frameType(), frame.frameFlags(), frame.framePhase(),
                     frame.frameRate(), frame.sampleStart(), frame.sampleEnd(), frame.sampleRate(), hex.empty() ? "-" : hex.c_str());
      }

      std::printf("DONE %zu %lu\n", frames.size(), samples.load());
      frames.clear();
   }

   std::fflush(stdout);
   command(decoderCommands, lab::RadioDecoderTask::Stop, json::object());
   executor.shutdown();

   return status;
}
