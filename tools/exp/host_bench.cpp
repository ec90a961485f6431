#include <stdio.h>
#include <chrono>
#include <vector>
typedef uint32_t u32; 
#include "../../tiny-ram-halo2_amd/csrc/field.h"
#include "../../tiny-ram-halo2_amd/csrc/hostcombine.h"
using namespace trh; using namespace trh::hostcombine;
int main(){
  // a random-ish point: take generator-like by doubling some XYZZ garbage that satisfies curve? Horner doesn't check curve membership; timing only
  std::vector<uint64_t> ws(52*16);
  uint64_t x=88172645463325252ull;
  for(auto&v:ws){x^=x<<13;x^=x>>7;x^=x<<17;v=x;} for(size_t i=3;i<ws.size();i+=4) ws[i]&=0x3fffffffffffffffull;
  auto t0=std::chrono::steady_clock::now();
  P acc; int reps=200;
  for(int r=0;r<reps;r++){ acc=horner<FpParams>(ws.data(),52,5); ws[0]^=acc.x.l[0]; }
  auto t1=std::chrono::steady_clock::now();
  printf("horner 52x5: %.1f us\n", std::chrono::duration<double,std::micro>(t1-t0).count()/reps);
  H a; memcpy(&a,ws.data(),32); a.l[3]&=0x3fffffffffffffffull;
  t0=std::chrono::steady_clock::now();
  for(int r=0;r<1000000;r++) a=mul<FpParams>(a,a);
  t1=std::chrono::steady_clock::now();
  printf("mul: %.1f ns (%llx)\n", std::chrono::duration<double,std::nano>(t1-t0).count()/1e6,(unsigned long long)a.l[0]);
  t0=std::chrono::steady_clock::now();
  for(int r=0;r<2000;r++) a=inv<FpParams>(a);
  t1=std::chrono::steady_clock::now();
  printf("inv: %.2f us\n", std::chrono::duration<double,std::micro>(t1-t0).count()/2000);
}
