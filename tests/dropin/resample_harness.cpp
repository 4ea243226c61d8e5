/*
 * TEST INFRASTRUCTURE — oracle for the adaptive resampler.
 *
 * Runs the reference's lab::SignalResamplingTask (compiled from /root/reference where it lies, unmodified) the way the
 * application does: submitted to an rt::Executor, fed SIGNAL_TYPE_RADIO_SAMPLES buffers on "radio.signal.raw",
 * observed on "adaptive.signal". The (value, offset) control points of every output buffer are written to a binary
 * file so that the GPU implementation (nfcgpu_resample_radio) can be compared bit for bit.
 *
 * usage: resample-ref in.wav out.bin [samples_per_buffer]
 * out.bin: per input buffer  u32 count_floats, then count_floats float32 values.
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
#include <vector>

#include <rt/Executor.h>
#include <rt/Logger.h>
#include <rt/Subject.h>

#include <hw/RecordDevice.h>
#include <hw/SignalBuffer.h>
#include <hw/SignalType.h>

#include <lab/tasks/SignalResamplingTask.h>

int main(int argc, char *argv[])
{
   if (argc < 3)
   {
      std::fprintf(stderr, "usage: %s in.wav out.bin [samples_per_buffer]\n", argv[0]);
      return 2;
   }

   const unsigned int perBuffer = argc > 3 ? (unsigned int)std::atoi(argv[3]) : 65536;

   rt::Logger::init(std::cerr);
   rt::Logger::setRootLevel(rt::Logger::WARN_LEVEL);

   std::mutex lock;
   std::vector<std::vector<float>> outputs;
   std::atomic<bool> finished {false};

   int status = 0;

   {
      rt::Executor executor(16, 4);

      executor.submit(lab::SignalResamplingTask::construct());

      auto *signal = rt::Subject<hw::SignalBuffer>::name("radio.signal.raw");
      auto *adaptive = rt::Subject<hw::SignalBuffer>::name("adaptive.signal");

      auto subscription = adaptive->subscribe([&](const hw::SignalBuffer &buffer) {
         if (!buffer.isValid())
         {
            finished = true;
            return;
         }

         std::lock_guard<std::mutex> guard(lock);
         outputs.emplace_back(buffer.data(), buffer.data() + buffer.limit());
      });

      std::this_thread::sleep_for(std::chrono::milliseconds(100));

      hw::RecordDevice source(argv[1]);

      if (!source.open(hw::RecordDevice::Mode::Read))
      {
         std::fprintf(stderr, "cannot open %s\n", argv[1]);
         status = 1;
      }
      else
      {
         const unsigned int sampleRate = std::get<unsigned int>(source.get(hw::SignalDevice::PARAM_SAMPLE_RATE));

         while (!source.isEof())
         {
            hw::SignalBuffer samples(perBuffer, 1, 1, sampleRate, 0, 0, hw::SignalType::SIGNAL_TYPE_RADIO_SAMPLES, 0);

            if (source.read(samples) > 0)
               signal->next(samples);
         }

         signal->next(hw::SignalBuffer());

         for (int i = 0; i < 60000 && !finished; i++)
            std::this_thread::sleep_for(std::chrono::milliseconds(5));

         if (!finished)
            status = 3;
      }

      executor.shutdown();
   }

   std::FILE *out = std::fopen(argv[2], "wb");

   if (!out)
      return 4;

   for (const auto &buffer: outputs)
   {
      const unsigned int count = (unsigned int)buffer.size();
      std::fwrite(&count, sizeof(count), 1, out);
      std::fwrite(buffer.data(), sizeof(float), count, out);
   }

   std::fclose(out);

   std::printf("DONE %s %zu buffers\n", argv[1], outputs.size());

   return status;
}
