/*
 * nfc_trace.hip — trace (.trz) writer behind the C ABI: decoded frames in the on-disk format the
 * reference application opens ("open trace") and tools/py_nfclab reads, so that what a GPU run decodes can be inspected
 * with the reference's own tools. Host code only (no kernel in this file).
 *
 * Format, as the reference's TraceStorageTask writes and reads it (lab-tasks/src/main/cpp/tasks/TraceStorageTask.cpp:
 * writeFrameEntry 461-520, readFrameEntry 380-449, the time range of writeFile 211-240): a gzip-compressed tar archive
 * with one entry "frame.json" = {"frames":[{...},...]}, per frame sampleStart, sampleEnd, sampleRate, timeStart, timeEnd,
 * techType, frameType, frameRate, frameFlags, framePhase, dateTime and, for frames with payload, frameData ("26",
 * "04:00": upper-case hex bytes separated by colons) and length. Times follow the decoder: timeStart = sampleStart /
 * sampleRate, dateTime = streamTime + timeStart (NfcA.cpp frame construction, NfcDecoder.cpp:449-463); a range
 * [rangeStart, rangeEnd] keeps the frames inside it and shifts their times and sample numbers to its start
 * (TraceStorageTask.cpp:461-483).
 *
 * The archive is written without a compression library: a ustar header + the entry + two zero blocks, wrapped in a gzip
 * member whose deflate stream consists of stored blocks (RFC 1951 3.2.4) - any inflate reads it, the reference's included.
 */
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nfcgpu.h"

namespace {

/* shortest decimal form that reads back as the same double (what nlohmann::json's dump and Python's repr give) */
void append_double(std::string &out, double v)
{
   char buf[40];

   for (int digits = 1; digits <= 17; digits++)
   {
      std::snprintf(buf, sizeof(buf), "%.*g", digits, v);

      double back = 0;
      if (std::sscanf(buf, "%lf", &back) == 1 && back == v)
         break;
   }

   out += buf;

   /* a JSON number that is a double keeps its fraction ("2.0", as the reference's writer prints it) */
   if (!std::strpbrk(buf, ".eEn"))
      out += ".0";
}

void append_uint(std::string &out, uint64_t v)
{
   char buf[24];
   std::snprintf(buf, sizeof(buf), "%llu", (unsigned long long)v);
   out += buf;
}

/* the table of CRC-32 (reflected 0xEDB88320), built once: a function-local static of class type is initialised exactly once
 * whatever threads call first (nfcgpu_trace_write_frames needs no context and may be called from several at a time) */
struct Crc32Table
{
   uint32_t entry[256];

   Crc32Table()
   {
      for (uint32_t i = 0; i < 256; i++)
      {
         uint32_t c = i;
         for (int k = 0; k < 8; k++)
            c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
         entry[i] = c;
      }
   }
};

uint32_t crc32_of(const uint8_t *p, size_t n)
{
   static const Crc32Table table;

   uint32_t c = 0xFFFFFFFFu;
   for (size_t i = 0; i < n; i++)
      c = table.entry[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
   return c ^ 0xFFFFFFFFu;
}

/* one ustar entry (name, 0664, regular file) followed by the archive's end marker */
std::vector<uint8_t> tar_of(const char *name, const std::string &content)
{
   std::vector<uint8_t> tar(512, 0);
   uint8_t *h = tar.data();

   std::snprintf((char *)h, 100, "%s", name);
   std::snprintf((char *)h + 100, 8, "%07o", 0664);
   std::snprintf((char *)h + 108, 8, "%07o", 0);
   std::snprintf((char *)h + 116, 8, "%07o", 0);
   std::snprintf((char *)h + 124, 12, "%011llo", (unsigned long long)content.size());
   std::snprintf((char *)h + 136, 12, "%011o", 0);
   std::memset(h + 148, ' ', 8);
   h[156] = '0';
   std::memcpy(h + 257, "ustar", 6);
   h[263] = '0';
   h[264] = '0';

   unsigned sum = 0;
   for (int i = 0; i < 512; i++)
      sum += h[i];
   std::snprintf((char *)h + 148, 8, "%06o", sum);
   h[154] = 0;
   h[155] = ' ';

   tar.insert(tar.end(), content.begin(), content.end());
   tar.resize((tar.size() + 511) / 512 * 512, 0);
   tar.resize(tar.size() + 1024, 0);
   return tar;
}

int write_gzip_stored(const char *path, const std::vector<uint8_t> &raw)
{
   std::FILE *f = std::fopen(path, "wb");
   if (!f)
      return NFCGPU_EIO;

   const uint8_t head[10] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 0xFF};
   bool ok = std::fwrite(head, 1, sizeof(head), f) == sizeof(head);

   size_t at = 0;
   do
   {
      const size_t n = raw.size() - at < 65535 ? raw.size() - at : 65535;
      const bool last = at + n == raw.size();
      const uint8_t block[5] = {(uint8_t)(last ? 1 : 0), (uint8_t)(n & 0xFF), (uint8_t)(n >> 8), (uint8_t)(~n & 0xFF), (uint8_t)((~n >> 8) & 0xFF)};
      ok = ok && std::fwrite(block, 1, 5, f) == 5 && (n == 0 || std::fwrite(raw.data() + at, 1, n, f) == n);
      at += n;
   } while (at < raw.size());

   const uint32_t crc = crc32_of(raw.data(), raw.size());
   const uint32_t len = (uint32_t)raw.size();
   const uint8_t tail[8] = {(uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24), (uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
   ok = ok && std::fwrite(tail, 1, 8, f) == 8;
   ok = (std::fclose(f) == 0) && ok;

   return ok ? NFCGPU_OK : NFCGPU_EIO;
}

}

extern "C" int nfcgpu_trace_write_frames(const char *path, const nfcgpu_frame *frames, uint32_t count, int64_t stream_time, double range_start, double range_end,
                                         uint32_t *written)
{
   if (!path || (count && !frames))
      return NFCGPU_EINVAL;

   const bool ranged = range_end > range_start; /* (0, 0: everything) */
   std::string json = "{\"frames\":[";
   uint32_t kept = 0;

   for (uint32_t i = 0; i < count; i++)
   {
      const nfcgpu_frame &f = frames[i];
      const double rate = (double)f.sample_rate;
      const double timeStart = rate > 0 ? (double)f.sample_start / rate : 0.0;
      const double timeEnd = rate > 0 ? (double)f.sample_end / rate : 0.0;

      if (ranged && (timeStart < range_start || timeEnd > range_end))
         continue;

      const double shift = ranged ? range_start : 0.0;
      const uint64_t offset = ranged ? (uint64_t)((double)f.sample_rate * range_start) : 0u;

      if (kept++)
         json += ",";

      /* (keys in alphabetical order, as nlohmann::json's object prints them) */
      json += "{\"dateTime\":";
      append_double(json, (double)stream_time + timeStart);
      if (f.length)
      {
         json += ",\"frameData\":\"";
         for (uint32_t b = 0; b < f.length && b < sizeof(f.data); b++)
         {
            char hex[4];
            std::snprintf(hex, sizeof(hex), b ? ":%02X" : "%02X", f.data[b]);
            json += hex;
         }
         json += "\"";
      }
      json += ",\"frameFlags\":";
      append_uint(json, f.frame_flags);
      json += ",\"framePhase\":";
      append_uint(json, f.frame_phase);
      json += ",\"frameRate\":";
      append_uint(json, f.frame_rate);
      json += ",\"frameType\":";
      append_uint(json, f.frame_type);
      if (f.length)
      {
         json += ",\"length\":";
         append_uint(json, f.length < sizeof(f.data) ? f.length : (uint32_t)sizeof(f.data)); /* (the bytes frameData holds) */
      }
      json += ",\"sampleEnd\":";
      append_uint(json, f.sample_end - offset);
      json += ",\"sampleRate\":";
      append_uint(json, f.sample_rate);
      json += ",\"sampleStart\":";
      append_uint(json, f.sample_start - offset);
      json += ",\"techType\":";
      append_uint(json, f.tech_type);
      json += ",\"timeEnd\":";
      append_double(json, timeEnd - shift);
      json += ",\"timeStart\":";
      append_double(json, timeStart - shift);
      json += "}";
   }

   json += "]}";

   if (written)
      *written = kept;

   return write_gzip_stored(path, tar_of("frame.json", json));
}
