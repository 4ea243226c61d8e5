/*
 * TEST INFRASTRUCTURE ONLY - an in-process stand-in for the handful of RCCL entry points the frame gather of
 * nfc-laboratory_amd/csrc/nfcgpu.hip uses (ncclAllGather, grouped ncclBroadcast): the ranks of a "communicator" are
 * threads of one process, device memory is plain memory (the emulated test build), a collective is a rendezvous of all
 * ranks followed by copies. Linked into tests/hostsim/libnfcgpu_emulated.so only and switched on by NFCGPU_FAKE_RCCL=1,
 * so that the rank logic of nfcgpu_gather_frames(_packed) - counts, the common verdict, offsets, ranks without records -
 * runs for 2 and 8 ranks on a box without GPUs (tests/test_gather_ranks_emulated.py). The real library looks up
 * librccl.so and knows nothing of this file.
 */
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct World
{
   int n = 0;
   std::mutex m;
   std::condition_variable cv;
   int arrived = 0;
   uint64_t generation = 0;
   std::vector<const void *> posted;

   /* every rank arrives, the last one wakes the others */
   void rendezvous()
   {
      std::unique_lock<std::mutex> lock(m);
      const uint64_t mine = generation;

      if (++arrived == n)
      {
         arrived = 0;
         generation++;
         cv.notify_all();
      }
      else
         cv.wait(lock, [&] { return generation != mine; });
   }
};

struct Comm
{
   std::shared_ptr<World> world;
   int rank;
};

std::mutex registryMutex;
std::map<uint64_t, std::weak_ptr<World>> registry;
uint64_t nextId = 1;

size_t width(int type)
{
   return type == 3 ? 4 : (type == 0 || type == 1 ? 1 : 4); /* ncclUint32 is all the gather uses */
}

}

/* ---- NFCGPU_FAKE_RCCL=shm (round 6): the ranks are PROCESSES, as they are under torch.distributed.run ----
 * The world is a POSIX shared-memory object named by the unique id: a header with a sense-reversing barrier on atomics and one
 * slot per rank through which a collective's bytes travel (pointers mean nothing across processes). Whoever joins first creates
 * it, the last to leave unlinks it. tests/test_bench_multi_rank_dry.py runs `bench.py --gpus 8` on it with eight processes. */
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int kShmMaxRanks = 16;
constexpr size_t kShmSlot = 8u << 20;

struct ShmHeader
{
   std::atomic<uint32_t> ready;   /* 1 once n has been set by the creator */
   std::atomic<uint32_t> arrived;
   std::atomic<uint32_t> generation;
   std::atomic<uint32_t> attached;
   uint32_t n;
   uint32_t pad[11];
};

struct ShmComm
{
   ShmHeader *h;
   char *slots;
   size_t bytes;
   int rank;
   char name[64];
};

bool shm_mode()
{
   const char *v = std::getenv("NFCGPU_FAKE_RCCL");
   return v && std::strcmp(v, "shm") == 0;
}

void shm_rendezvous(ShmHeader *h)
{
   const uint32_t mine = h->generation.load(std::memory_order_acquire);

   if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == h->n)
   {
      h->arrived.store(0, std::memory_order_relaxed);
      h->generation.fetch_add(1, std::memory_order_acq_rel);
   }
   else
   {
      /* (ranks may outnumber the processors: yield, then sleep) */
      for (uint32_t spins = 0; h->generation.load(std::memory_order_acquire) == mine; spins++)
      {
         if (spins < 64)
            sched_yield();
         else
            usleep(200);
      }
   }
}

int shm_init(void **comm, int nRanks, uint64_t key, int rank)
{
   if (nRanks < 1 || nRanks > kShmMaxRanks || rank < 0 || rank >= nRanks)
      return 1;

   ShmComm *c = new ShmComm();
   std::snprintf(c->name, sizeof(c->name), "/nfcfake_%016llx", (unsigned long long)key);
   c->bytes = sizeof(ShmHeader) + (size_t)kShmMaxRanks * kShmSlot;
   c->rank = rank;

   bool creator = true;
   int fd = shm_open(c->name, O_RDWR | O_CREAT | O_EXCL, 0600);
   if (fd < 0 && errno == EEXIST)
   {
      creator = false;
      fd = shm_open(c->name, O_RDWR, 0600);
   }
   if (fd < 0)
   {
      delete c;
      return 1;
   }
   if (creator && ftruncate(fd, (off_t)c->bytes) != 0)
   {
      close(fd);
      shm_unlink(c->name);
      delete c;
      return 1;
   }
   if (!creator)
   {
      /* (the creator may not have sized it yet) */
      struct stat st;
      for (int tries = 0; tries < 20000; tries++)
      {
         if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->bytes)
            break;
         usleep(500);
      }
   }

   void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
   close(fd);
   if (p == MAP_FAILED)
   {
      delete c;
      return 1;
   }

   c->h = (ShmHeader *)p;
   c->slots = (char *)p + sizeof(ShmHeader);

   if (creator)
   {
      c->h->n = (uint32_t)nRanks;
      c->h->ready.store(1, std::memory_order_release);
   }
   else
   {
      while (c->h->ready.load(std::memory_order_acquire) == 0)
         usleep(200);
      if (c->h->n != (uint32_t)nRanks)
      {
         munmap(p, c->bytes);
         delete c;
         return 1;
      }
   }

   c->h->attached.fetch_add(1, std::memory_order_acq_rel);
   *comm = c;
   shm_rendezvous(c->h); /* like the real call: returns once every rank has joined */
   return 0;
}

/* moves `bytes` from every rank (or from `root` only) through the slots, a slot's worth at a time */
void shm_exchange(ShmComm *c, const void *send, void *recv, size_t bytes, int root)
{
   const uint32_t n = c->h->n;

   for (size_t done = 0; done < bytes || (bytes == 0 && done == 0); done += kShmSlot)
   {
      const size_t part = bytes - done < kShmSlot ? bytes - done : kShmSlot;

      if (root < 0 || c->rank == root)
         std::memcpy(c->slots + (size_t)c->rank * kShmSlot, (const char *)send + done, part);
      shm_rendezvous(c->h);
      if (root < 0)
      {
         for (uint32_t i = 0; i < n; i++)
            std::memcpy((char *)recv + (size_t)i * bytes + done, c->slots + (size_t)i * kShmSlot, part);
      }
      else
         std::memcpy((char *)recv + done, c->slots + (size_t)root * kShmSlot, part);
      shm_rendezvous(c->h);

      if (bytes == 0)
         break;
   }
}

}

extern "C" {

struct FakeNcclId
{
   char internal[128];
};

int fake_ncclGetUniqueId(void *id)
{
   if (shm_mode())
   {
      std::memset(id, 0, 128);
      struct timespec ts;
      clock_gettime(CLOCK_REALTIME, &ts);
      static std::atomic<uint32_t> serial {0};
      const uint64_t v = ((uint64_t)getpid() << 40) ^ ((uint64_t)ts.tv_sec << 20) ^ (uint64_t)ts.tv_nsec ^ ((uint64_t)serial.fetch_add(1) << 56) ^ 0x5a5a000000000001ull;
      std::memcpy(id, &v, 8);
      return 0;
   }

   std::lock_guard<std::mutex> lock(registryMutex);
   std::memset(id, 0, 128);
   const uint64_t v = nextId++;
   std::memcpy(id, &v, 8);
   return 0;
}

int fake_ncclCommInitRank(void **comm, int nRanks, FakeNcclId id, int rank)
{
   uint64_t key = 0;
   std::memcpy(&key, id.internal, 8);

   if (shm_mode())
      return shm_init(comm, nRanks, key, rank);

   std::shared_ptr<World> world;
   {
      std::lock_guard<std::mutex> lock(registryMutex);
      world = registry[key].lock();
      if (!world)
      {
         world = std::make_shared<World>();
         world->n = nRanks;
         world->posted.resize(nRanks);
         registry[key] = world;
      }
   }

   if (world->n != nRanks || rank < 0 || rank >= nRanks)
      return 1;

   *comm = new Comm {world, rank};
   world->rendezvous(); /* like the real call: returns once every rank has joined */
   return 0;
}

int fake_ncclCommDestroy(void *comm)
{
   if (shm_mode())
   {
      ShmComm *c = (ShmComm *)comm;
      const bool last = c->h->attached.fetch_sub(1, std::memory_order_acq_rel) == 1;
      munmap((void *)c->h, c->bytes);
      if (last)
         shm_unlink(c->name);
      delete c;
      return 0;
   }

   delete (Comm *)comm;
   return 0;
}

int fake_ncclAllGather(const void *send, void *recv, size_t count, int type, void *comm, void *)
{
   if (shm_mode())
   {
      shm_exchange((ShmComm *)comm, send, recv, count * width(type), -1);
      return 0;
   }

   Comm *c = (Comm *)comm;
   World &w = *c->world;
   const size_t bytes = count * width(type);

   w.posted[c->rank] = send;
   w.rendezvous();
   for (int i = 0; i < w.n; i++)
      std::memmove((char *)recv + (size_t)i * bytes, w.posted[i], bytes);
   w.rendezvous(); /* nobody's send buffer changes while another rank still reads it */
   return 0;
}

int fake_ncclBroadcast(const void *send, void *recv, size_t count, int type, int root, void *comm, void *)
{
   if (shm_mode())
   {
      shm_exchange((ShmComm *)comm, send, recv, count * width(type), root);
      return 0;
   }

   Comm *c = (Comm *)comm;
   World &w = *c->world;

   if (c->rank == root)
      w.posted[root] = send;
   w.rendezvous();
   std::memmove(recv, w.posted[root], count * width(type));
   w.rendezvous();
   return 0;
}

/* (every rank issues the same broadcasts in the same order: they can simply run as they come) */
int fake_ncclGroupStart() { return 0; }
int fake_ncclGroupEnd() { return 0; }

}
