/*
 * TEST INFRASTRUCTURE - the reference's lab::TraceStorageTask (the task behind "open trace" / "save trace" of the Qt
 * application, compiled from /root/reference where it lies) driven through its subjects, as the check of the .trz
 * traces nfc-laboratory_amd/trz.py writes from batch output:
 *
 *   trace-ref read  file.trz              the task reads the trace (storage.command Read) and publishes its frames on
 *                                         storage.frame; they are printed one per line
 *   trace-ref write file.trz frames.txt   the frames of frames.txt are published on radio.decoder.frame, then the task
 *                                         writes them (storage.command Write): the reference's own rendering of those
 *                                         frames, to compare trz.py's with
 * frame line:  tech type flags phase rate sampleStart sampleEnd sampleRate timeStart timeEnd dateTime hexdata|-
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <list>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>

#include <nlohmann/json.hpp>

#include <rt/Event.h>
#include <rt/Executor.h>
#include <rt/Logger.h>
#include <rt/Subject.h>

#include <lab/data/RawFrame.h>
#include <lab/tasks/TraceStorageTask.h>

using json = nlohmann::json;

static bool command(rt::Subject<rt::Event> *subject, int code, const json &data)
{
   std::atomic<int> outcome {0};

   subject->next({code, [&outcome] { outcome = 1; }, [&outcome](int, const std::string &) { outcome = -1; }, {{"data", data.dump()}}});

   for (int i = 0; i < 6000 && outcome == 0; i++)
      std::this_thread::sleep_for(std::chrono::milliseconds(5));

   return outcome == 1;
}

int main(int argc, char *argv[])
{
   if (argc < 3)
      return 2;

   rt::Logger::init(std::cerr);
   rt::Logger::setRootLevel(rt::Logger::WARN_LEVEL);

   rt::Executor executor(16, 4);
   executor.submit(lab::TraceStorageTask::construct());

   auto *commands = rt::Subject<rt::Event>::name("storage.command");
   auto *stored = rt::Subject<lab::RawFrame>::name("storage.frame");
   auto *decoded = rt::Subject<lab::RawFrame>::name("radio.decoder.frame");

   std::mutex lock;
   std::list<lab::RawFrame> frames;

   auto subscription = stored->subscribe([&](const lab::RawFrame &frame) {
      if (frame.isValid())
      {
         std::lock_guard<std::mutex> guard(lock);
         frames.push_back(frame);
      }
   });

   std::this_thread::sleep_for(std::chrono::milliseconds(100));

   const std::string mode = argv[1];
   int status = 0;

   if (mode == "read")
   {
      if (!command(commands, lab::TraceStorageTask::Read, {{"fileName", argv[2]}}))
      {
         std::fprintf(stderr, "read command rejected\n");
         status = 1;
      }

      std::lock_guard<std::mutex> guard(lock);

      for (const lab::RawFrame &f: frames)
      {
         std::printf("%u %u %u %u %u %lu %lu %lu %.9f %.9f %.9f ", f.techType(), f.frameType(), f.frameFlags(), f.framePhase(), f.frameRate(),
                     (unsigned long)f.sampleStart(), (unsigned long)f.sampleEnd(), (unsigned long)f.sampleRate(), f.timeStart(), f.timeEnd(),
                     f.dateTime());
         if (f.limit() == 0)
            std::printf("-");
         for (unsigned int i = 0; i < f.limit(); i++)
            std::printf("%02x", (unsigned)f[i]);
         std::printf("\n");
      }
   }
   else if (mode == "write" && argc >= 4)
   {
      std::ifstream in(argv[3]);
      std::string line;
      double last = 0;

      while (std::getline(in, line))
      {
         std::istringstream w(line);
         unsigned int tech, type, flags, phase, rate;
         unsigned long start, end, fs;
         double t0, t1, date;
         std::string hex;

         if (!(w >> tech >> type >> flags >> phase >> rate >> start >> end >> fs >> t0 >> t1 >> date >> hex))
            continue;

         lab::RawFrame frame(tech, type);
         frame.setFrameFlags(flags);
         frame.setFramePhase(phase);
         frame.setFrameRate(rate);
         frame.setSampleStart(start);
         frame.setSampleEnd(end);
         frame.setSampleRate(fs);
         frame.setTimeStart(t0);
         frame.setTimeEnd(t1);
         frame.setDateTime(date);

         for (size_t i = 0; hex != "-" && i + 1 < hex.size(); i += 2)
            frame.put((unsigned char)std::stoi(hex.substr(i, 2), nullptr, 16));

         frame.flip();
         decoded->next(frame);
         last = t1 > last ? t1 : last;
      }

      if (!command(commands, lab::TraceStorageTask::Write, {{"fileName", argv[2]}, {"timeStart", 0.0}, {"timeEnd", last + 1.0}}))
      {
         std::fprintf(stderr, "write command rejected\n");
         status = 1;
      }
   }
   else
   {
      status = 2;
   }

   std::fflush(stdout);
   executor.shutdown();
   frames.clear();

   return status;
}
