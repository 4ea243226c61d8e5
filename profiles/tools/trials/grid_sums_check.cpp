/* DESIGN CHECK, CPU only, test infrastructure: are the search bank's running sums exact box sums on int16-grid input?
 * Steps the product's device step machine over a capture (raw float32 file) up to its first lock and compares the running
 * sum of each NFC-A correlator with the box sum of the same window taken from a double-precision prefix sum, up to the
 * constant the sum starts with (the detectors are first stepped at clock 1024 with an empty accumulator).
 *   g++ -std=c++17 -O2 -ffp-contract=off -msse3 -mno-avx -I../../../nfc-laboratory_amd/csrc grid_sums_check.cpp -o /tmp/grid_sums_check
 * Round-1 result: 0 differences in 507 237 comparisons on four fixtures (int16 grid); 185 356 of 185 775 differ on a fuzzed
 * capture (general fp32). The first is what a prefix-sum search kernel for grid input rests on. */
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
#define NFC_DEV static inline
static inline uint32_t a_(uint32_t*p,uint32_t v){uint32_t o=*p;*p+=v;return o;}
#define NFC_ATOMIC_ADD(p,v) a_((p),(v))
#define NFC_ANY(x) (x)
#include "nfc_core.hpp"
#include "nfc_config.hpp"
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); std::vector<float> x; float v; while(fread(&v,4,1,f)==1) x.push_back(v); fclose(f);
  NfcHostParams p; p.sampleRate=10000000; p.enabled=0xF; NfcConfig cfg; nfc_build_config(p,cfg);
  std::vector<float> rings((4*NFC_HIST+NFC_PROD+cfg.corrTotal)*NFC_LANES,0.f); std::vector<uint8_t> bytes(NFC_STREAM_BYTES,0); std::vector<uint32_t> arena(1u<<22,0);
  NfcLaneMem mem; mem.linked=false; mem.flags=nullptr; mem.ring=rings.data(); mem.lane=0; mem.exact=true; mem.bytes=bytes.data(); uint32_t ctl[2]={0,0}; mem.sink=arena.data(); mem.sinkCursor=&ctl[0]; mem.sinkDropped=&ctl[1]; mem.sinkWords=arena.size(); mem.streamId=0;
  NfcStreamState s; NfcStreamCold cold; memset(&s,0,sizeof s); memset(&cold,0,sizeof cold); mem.cold=&cold; mem.tables=&cfg; nfc_state_init(cfg,s,cold,false);
  uint64_t checked=0, bad=0, gapless=1, locks=0; double off[3]; bool have[3]={false,false,false};
  std::vector<double> pre(x.size()+1,0.0); for(size_t i=0;i<x.size();i++) pre[i+1]=pre[i]+(double)x[i];
  for(size_t n=0;n<x.size();n++){
    uint32_t before=s.lockTech;
    nfc_step(cfg,s,mem,x[n],true);
    if(s.lockTech&&!before){locks++; gapless=0;}
    if(!gapless) break;               // only the gapless prefix: afterwards the sums carry the bookkeeping of the gap
    if(s.clock>=1024 && s.lockTech==0 && !(s.env<cfg.powerThreshold)){
      for(int r=0;r<3;r++){
        long d=cfg.a[r].delay, w=cfg.a[r].p2; long hi=(long)n-d, lo=hi-w;  // window (lo, hi]
        if(lo< (long)1024) continue;   // the bank only runs from clock 1024 on: sums started there
        // the sum runs since the bank was first stepped at clock 1024: box sum only once the window lies after that
        double box=pre[hi+1]-pre[lo+1];
        if(!have[r]){have[r]=true; off[r]=(double)s.u.search.detA[r].acc-box;} checked++; if((double)s.u.search.detA[r].acc-box!=off[r]){ if(bad<5) printf("n=%zu r=%d acc=%.9g box=%.9g\n",n,r,s.u.search.detA[r].acc,box); bad++; }
      }
    }
  }
  printf("checked %llu mismatching %llu (gapless prefix, first lock after %llu locks)\n",(unsigned long long)checked,(unsigned long long)bad,(unsigned long long)locks);
}
