/*
 * TEST INFRASTRUCTURE ONLY - a wavefront on the CPU: the 64 lanes of a wave are 64 fibres that run the same function
 * and meet at barriers. With it the text of a wave-cooperative kernel (nfc-laboratory_amd/csrc/nfc_wave.hpp) runs on a
 * box without a GPU, cross-lane operations included: a ballot or an LDS hand-over is "every lane deposits, barrier,
 * every lane reads". Lanes run one after the other between barriers, so code that reads LDS another lane wrote without a
 * barrier in between reads the old value here: stricter than the hardware, which is what a test wants.
 * x86-64 only (a 12-instruction context switch below).
 */
#ifndef NFC_WAVESIM_HPP
#define NFC_WAVESIM_HPP

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace wavesim {

constexpr int kLanes = 64;
constexpr size_t kStack = 512 * 1024;

extern "C" void wavesim_switch(void **from, void *to);

asm(R"(
.text
.globl wavesim_switch
.type wavesim_switch,@function
wavesim_switch:
   pushq %rbp
   pushq %rbx
   pushq %r12
   pushq %r13
   pushq %r14
   pushq %r15
   movq %rsp, (%rdi)
   movq %rsi, %rsp
   popq %r15
   popq %r14
   popq %r13
   popq %r12
   popq %rbx
   popq %rbp
   ret
.size wavesim_switch,.-wavesim_switch
)");

struct Wave
{
   void *sp[kLanes];
   bool done[kLanes];
   uint32_t barriers[kLanes];
   void *scheduler = nullptr;
   int current = 0;
   void (*body)(void *) = nullptr;
   void *arg = nullptr;
   char *stacks = nullptr;
   uint64_t word[kLanes];
   unsigned char record[8192];
};

inline Wave *&cur()
{
   static thread_local Wave *w = nullptr;
   return w;
}

inline uint32_t lane() { return (uint32_t)cur()->current; }

inline void barrier()
{
   Wave *w = cur();
   w->barriers[w->current]++;
   wavesim_switch(&w->sp[w->current], w->scheduler);
}

inline void trampoline()
{
   Wave *w = cur();
   w->body(w->arg);
   w->done[w->current] = true;
   wavesim_switch(&w->sp[w->current], w->scheduler);
   std::abort(); /* a finished fibre is never resumed */
}

/* run body(arg) in 64 lanes */
inline void run(void (*body)(void *), void *arg)
{
   static thread_local Wave wave;
   Wave *w = &wave;

   if (!w->stacks)
      w->stacks = (char *)std::malloc(kStack * kLanes);

   w->body = body;
   w->arg = arg;
   cur() = w;

   for (int l = 0; l < kLanes; l++)
   {
      char *top = w->stacks + kStack * (size_t)(l + 1);
      top = (char *)((uintptr_t)top & ~(uintptr_t)15);
      void **sp = (void **)top;
      *--sp = nullptr;                 /* (return address slot of the trampoline: keeps the entry alignment of a call) */
      *--sp = (void *)&trampoline;
      for (int r = 0; r < 6; r++)
         *--sp = nullptr;
      w->sp[l] = sp;
      w->done[l] = false;
      w->barriers[l] = 0;
   }

   int remaining = kLanes;

   while (remaining)
   {
      uint32_t epoch = 0;
      bool first = true;

      for (int l = 0; l < kLanes; l++)
      {
         if (w->done[l])
            continue;

         w->current = l;
         wavesim_switch(&w->scheduler, w->sp[l]);

         if (w->done[l])
            remaining--;
         else
         {
            /* every lane still running must be at the same barrier */
            if (first)
            {
               epoch = w->barriers[l];
               first = false;
            }
            else if (w->barriers[l] != epoch)
            {
               std::fprintf(stderr, "wavesim: lanes at different barriers (lane %d: %u, expected %u)\n", l, w->barriers[l], epoch);
               std::abort();
            }
         }
      }
   }

   cur() = nullptr;
}

inline uint64_t ballot(bool p)
{
   Wave *w = cur();
   w->word[w->current] = p ? 1u : 0u;
   barrier();
   uint64_t m = 0;
   for (int l = 0; l < kLanes; l++)
      m |= (uint64_t)(w->word[l] & 1u) << l;
   barrier();
   return m;
}

template <class T> inline T shfl(T v, uint32_t src)
{
   static_assert(sizeof(T) <= 8, "shfl of at most 8 bytes");
   Wave *w = cur();
   uint64_t bits = 0;
   std::memcpy(&bits, &v, sizeof(T));
   w->word[w->current] = bits;
   barrier();
   const uint64_t got = w->word[src & 63u];
   barrier();
   T r;
   std::memcpy(&r, &got, sizeof(T));
   return r;
}

/* end of a uniform region that only lane 0 executed: its record goes to every lane */
template <class U> inline void uniform_sync(U &u)
{
   static_assert(sizeof(U) <= sizeof(Wave::record), "uniform record too large");
   Wave *w = cur();
   if (w->current == 0)
      std::memcpy(w->record, &u, sizeof(U));
   barrier();
   if (w->current != 0)
      std::memcpy(&u, w->record, sizeof(U));
   barrier();
}

} // namespace wavesim

#endif
