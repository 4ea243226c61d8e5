/*
 * TEST INFRASTRUCTURE. Drives one lab::NfcDecoder through its public interface (the drop-in seam) with a
 * scripted sequence of calls and prints what comes back. Linked twice from this one source: with the reference's
 * decoder (oracle/_ref/api-ref) and with the lab::NfcDecoder shim on libnfcgpu.so (oracle/_ref/api-gpu); the outputs
 * of the two on the same script must be identical (tests/test_interface_sequences.py). Unlike a capture replay this
 * exercises the interface semantics: setters between buffers, initialize() in mid-stream, sample-rate changes, invalid
 * buffers, empty buffers, technologies switched off while locked.
 *
 *   api_harness samples.f32 script.txt
 * script lines:  [@k] enable <A|B|F|V> <0|1> | power <f> | corr <A|B|F|V> <f> | depth <A|B|F|V> <min> <max> | rate <hz>
 *                     time <t> | init | feed <first> <count> <hz> | invalid | drop
 * @k selects decoder k (created at its first line, default 0): several decoders live side by side, as several
 * RadioDecoderTasks would (with the shim they share one GPU context); drop destroys decoder k.
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <hw/SignalType.h>
#include <hw/SignalBuffer.h>
#include <lab/data/RawFrame.h>
#include <lab/nfc/NfcDecoder.h>

#ifdef NFC_DEFINED_FRAME_STORAGE
/* Reference build only (oracle/build_ref.sh): frame storage cleared and never recycled, so that what the reference reads
 * beyond truncated frames is defined (zero) instead of a leftover of an earlier frame; see oracle/ref_capi.cpp. */
#include <cstring>
extern "C" int __real_posix_memalign(void **ptr, size_t alignment, size_t size);
extern "C" int __wrap_posix_memalign(void **ptr, size_t alignment, size_t size)
{
   int res = __real_posix_memalign(ptr, alignment, size);
   if (res == 0)
      std::memset(*ptr, 0, size);
   return res;
}
static std::list<std::list<lab::RawFrame>> kept;
#define KEEP(frames) kept.push_back(frames)
#else
#define KEEP(frames) (void)0
#endif

static void print(const char *tag, const std::list<lab::RawFrame> &frames)
{
   for (const lab::RawFrame &f: frames)
   {
      std::printf("%s %u %u %u %u %u %lu %lu %lu %.9f %.9f %.3f ", tag, f.techType(), f.frameType(), f.frameFlags(), f.framePhase(),
                  f.frameRate(), (unsigned long)f.sampleStart(), (unsigned long)f.sampleEnd(), (unsigned long)f.sampleRate(), f.timeStart(),
                  f.timeEnd(), f.dateTime());
      for (unsigned int i = 0; i < f.limit(); i++)
         std::printf("%02x", (unsigned)f[i]);
      std::printf("\n");
   }
}

int main(int argc, char *argv[])
{
   if (argc != 3)
      return 2;

   std::ifstream raw(argv[1], std::ios::binary);
   std::vector<char> bytes((std::istreambuf_iterator<char>(raw)), std::istreambuf_iterator<char>());
   const float *samples = reinterpret_cast<const float *>(bytes.data());
   const size_t total = bytes.size() / sizeof(float);

   std::map<int, std::unique_ptr<lab::NfcDecoder>> decoders;
   std::ifstream script(argv[2]);
   std::string line;
   int step = 0;

   while (std::getline(script, line))
   {
      std::istringstream in(line);
      std::string op, tech;
      in >> op;
      step++;

      int k = 0;
      if (!op.empty() && op[0] == '@')
      {
         k = std::atoi(op.c_str() + 1);
         in >> op;
      }

      if (op == "drop")
      {
         decoders.erase(k);
         std::printf("S%d @%d dropped\n", step, k);
         continue;
      }

      if (!decoders.count(k))
         decoders[k].reset(new lab::NfcDecoder());

      lab::NfcDecoder &decoder = *decoders[k];

      auto which = [&]() { in >> tech; return tech.empty() ? 'A' : tech[0]; };
      auto number = [&]() { std::string t; in >> t; return std::strtof(t.c_str(), nullptr); }; /* accepts nan */

      if (op == "enable")
      {
         char t = which(); int on = 0; in >> on;
         if (t == 'A') decoder.setEnableNfcA(on); if (t == 'B') decoder.setEnableNfcB(on);
         if (t == 'F') decoder.setEnableNfcF(on); if (t == 'V') decoder.setEnableNfcV(on);
      }
      else if (op == "power") { float v = number(); decoder.setPowerLevelThreshold(v); }
      else if (op == "corr")
      {
         char t = which(); float v = number();
         if (t == 'A') decoder.setCorrelationThresholdNfcA(v); if (t == 'B') decoder.setCorrelationThresholdNfcB(v);
         if (t == 'F') decoder.setCorrelationThresholdNfcF(v); if (t == 'V') decoder.setCorrelationThresholdNfcV(v);
      }
      else if (op == "depth")
      {
         char t = which(); float lo = number(), hi = number();
         if (t == 'A') decoder.setModulationThresholdNfcA(lo, hi); if (t == 'B') decoder.setModulationThresholdNfcB(lo, hi);
         if (t == 'F') decoder.setModulationThresholdNfcF(lo, hi); if (t == 'V') decoder.setModulationThresholdNfcV(lo, hi);
      }
      else if (op == "rate") { long hz = 0; in >> hz; decoder.setSampleRate(hz); }
      else if (op == "time") { long t = 0; in >> t; decoder.setStreamTime(t); }
      else if (op == "init") { decoder.initialize(); }
      else if (op == "feed")
      {
         size_t first = 0, count = 0; long hz = 0;
         in >> first >> count >> hz;
         if (first > total) first = total;
         if (first + count > total) count = total - first;
         hw::SignalBuffer buffer(count ? count : 1, 1, 1, hz, 0, 0, hw::SignalType::SIGNAL_TYPE_RADIO_SAMPLES, 0);
         buffer.put(samples + first, count).flip();
         char tag[32];
         std::snprintf(tag, sizeof(tag), "F%d", step);
         std::list<lab::RawFrame> frames = decoder.nextFrames(buffer);
         print(tag, frames);
         KEEP(frames);
      }
      else if (op == "invalid")
      {
         hw::SignalBuffer invalid;
         char tag[32];
         std::snprintf(tag, sizeof(tag), "I%d", step);
         std::list<lab::RawFrame> frames = decoder.nextFrames(invalid);
         print(tag, frames);
         KEEP(frames);
      }

      std::printf("S%d @%d rate=%ld time=%ld power=%.6f A=%d B=%d F=%d V=%d corr=%.6f/%.6f/%.6f/%.6f depth=%.6f-%.6f/%.6f-%.6f/%.6f-%.6f/%.6f-%.6f debug=%d\n",
                  step, k, decoder.sampleRate(), decoder.streamTime(),
                  decoder.powerLevelThreshold(), (int)decoder.isNfcAEnabled(), (int)decoder.isNfcBEnabled(), (int)decoder.isNfcFEnabled(),
                  (int)decoder.isNfcVEnabled(), decoder.correlationThresholdNfcA(), decoder.correlationThresholdNfcB(),
                  decoder.correlationThresholdNfcF(), decoder.correlationThresholdNfcV(), decoder.modulationThresholdNfcAMin(),
                  decoder.modulationThresholdNfcAMax(), decoder.modulationThresholdNfcBMin(), decoder.modulationThresholdNfcBMax(),
                  decoder.modulationThresholdNfcFMin(), decoder.modulationThresholdNfcFMax(), decoder.modulationThresholdNfcVMin(),
                  decoder.modulationThresholdNfcVMax(), (int)decoder.isDebugEnabled());
   }

   return 0;
}
