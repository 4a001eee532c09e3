/* Test infrastructure: eg_exp / eg_log / eg_pow of include/egdst_math.h against the platform libm, bit for bit.
 * Built and run by tests/test_math_vs_libm.py:  gcc -O2 [-mfma] -ffp-contract=off math_check.c -lm;  ./a.out N
 * Argument mixes: the ranges the models use (utility, shocks, transforms), wide ranges, raw bit patterns (nan, inf,
 * subnormals, negative bases), the overflow/underflow borders, and a grid of special values. */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include "egdst_math.h"
static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t rnd(void){ uint64_t s1=s[0], s0=s[1]; s[0]=s0; s1^=s1<<23; s[1]=s1^s0^(s1>>18)^(s0>>5); return s[1]+s0; }
static double u01(void){ return (rnd()>>11) * 0x1p-53; }
static int same(double a,double b){ uint64_t x,y; memcpy(&x,&a,8); memcpy(&y,&b,8); return x==y || (a!=a && b!=b); }
int main(int argc,char**argv){
  long n = argc>1? atol(argv[1]) : 10000000; long bad[3]={0,0,0};
  for(long i=0;i<n;i++){
    double x;
    switch(i&7){ case 0: x=(u01()-0.5)*20; break; case 1: x=(u01()-0.5)*1500; break; case 2: x=(u01()-0.5)*2; break;
      case 3: x=(u01()-0.5)*1e-3; break; case 4: {uint64_t b=rnd(); memcpy(&x,&b,8);} break; case 5: x=-700-u01()*50; break; case 6: x=700+u01()*12; break; default: x=(u01()-0.5)*200; }
    if(!same(eg_exp(x),exp(x))){ if(bad[0]++<5) printf("exp(%a) %a vs %a\n",x,eg_exp(x),exp(x)); }
    double p;
    switch(i&7){ case 0: p=u01()*100; break; case 1: p=0.9+u01()*0.2; break; case 2: p=exp((u01()-0.5)*1400); break; case 3: p=u01(); break;
      case 4: {uint64_t b=rnd(); memcpy(&p,&b,8);} break; case 5: p=u01()*0x1p-1030; break; case 6: p=1+ (u01()-0.5)*1e-6; break; default: p=u01()*1e6; }
    if(!same(eg_log(p),log(p))){ if(bad[1]++<5) printf("log(%a) %a vs %a\n",p,eg_log(p),log(p)); }
    double a,b;
    switch(i&7){ case 0: a=u01()*100; b=(u01()-0.5)*8; break; case 1: a=u01()*50; b=1-u01()*3; break; case 2: a=exp((u01()-0.5)*100); b=(u01()-0.5)*40; break;
      case 3: a=u01()*10; b=-1.5; break; case 4: {uint64_t t=rnd(); memcpy(&a,&t,8); t=rnd(); memcpy(&b,&t,8);} break; case 5: a=-u01()*10; b=(double)((long)(rnd()%21)-10); break;
      case 6: a=u01()*0x1p-1030; b=(u01()-0.5)*2; break; default: a=1+(u01()-0.5)*1e-3; b=(u01()-0.5)*1e6; }
    if(!same(eg_pow(a,b),pow(a,b))){ if(bad[2]++<5) printf("pow(%a,%a) %a vs %a\n",a,b,eg_pow(a,b),pow(a,b)); }
  }
  double sp[]={0.0,-0.0,1.0,-1.0,INFINITY,-INFINITY,NAN,0.5,2.0,3.0,-3.0,-2.0,0x1p-1074,0x1p1023,1e300,-1e300,709.78,-745.13,-745.14,710.0,1e-300, 0x1p-1022, 4.0, 5.0, 1e16, 9.007199254740993e15};
  int ns=sizeof sp/sizeof *sp;
  for(int i=0;i<ns;i++){ if(!same(eg_exp(sp[i]),exp(sp[i]))){bad[0]++; printf("exp sp %a\n",sp[i]);} if(!same(eg_log(sp[i]),log(sp[i]))){bad[1]++; printf("log sp %a: %a vs %a\n",sp[i],eg_log(sp[i]),log(sp[i]));}
    for(int j=0;j<ns;j++) if(!same(eg_pow(sp[i],sp[j]),pow(sp[i],sp[j]))){bad[2]++; printf("pow sp %a %a: %a vs %a\n",sp[i],sp[j],eg_pow(sp[i],sp[j]),pow(sp[i],sp[j]));} }
  printf("n=%ld mismatches exp=%ld log=%ld pow=%ld\n",n,bad[0],bad[1],bad[2]);
  return (bad[0]||bad[1]||bad[2])?1:0; }
