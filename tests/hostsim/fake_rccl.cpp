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

extern "C" {

struct FakeNcclId
{
   char internal[128];
};

int fake_ncclGetUniqueId(void *id)
{
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
   delete (Comm *)comm;
   return 0;
}

int fake_ncclAllGather(const void *send, void *recv, size_t count, int type, void *comm, void *)
{
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
