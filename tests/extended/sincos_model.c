/*
 * TEST INFRASTRUCTURE — CPU model of the device's fast paths (doppler_amd/csrc/dpx_sincos.h: the two argument reductions
 * of round 4, reduce_small / reduce_large_quick, and round 5's nine-operation polynomials sincos_horner /
 * sincos_horner_quadrants with the two-term reduce_large_two), enumerated against the oracle's restated glibc 2.35 sincosf
 * (oracle/sincosf_glibc.c, itself pinned to libm over all 2^32 arguments).  --v 1 (the FMA build) models the round-5 forms,
 * --v 0 (the SSE2 build) glibc's own polynomial order on the round-4 reductions: exactly what the device executes for each.
 *
 * The device code replaces glibc's operation sequence in two places by a cheaper one that is claimed to give the
 * same two floats for every argument of its range.  That claim is a finite statement; this program checks it
 * argument by argument with the same IEEE double operations the device executes (fma, add, mul, conversion):
 *     sincos_model [--v 0|1] [--plain] [--lo BITS] [--hi BITS]
 *       default: the large range, |y| in [120, 2^29), both signs: quick reduction vs the integer path
 *       --mixed: |y| in [0, 2^29), the path of wavefronts whose lanes lie in both fast ranges (sincosf_mixed; FMA build)
 *       --plain: |y| in [2^-12, 120): rounding by fused multiply-add against 1.5*2^52 vs glibc's truncating conversion
 *       --v:     libm build (1 = FMA contraction, 0 = SSE2)
 * Exit status 0 iff there is no mismatch.  tests/test_oracle.py runs all four enumerations (about 20 s on 8 cores);
 * the device itself is enumerated by tests/extended/exhaustive_device_sincos.py (profiles/r04_exhaustive_device_sincos.json).
 * Build: gcc -O2 -mfma -ffp-contract=off -Ioracle -o sincos_model tests/extended/sincos_model.c oracle/sincosf_glibc.c -lm -lpthread
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sincosf_glibc.h"
static inline uint32_t fb(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline uint64_t db(double f){uint64_t u;memcpy(&u,&f,8);return u;}
static inline double mad(int v,double a,double b,double c){ if(v) return __builtin_fma(a,b,c); double p=a*b; return p+c; }
static const double MAGIC=0x1.8p52, C=0x1.45f306dc9c883p-1,c1=0x1.45f306d000000p-1,c2=0x1.9391054a7f09dp-30,c3=0x1.7d1f534ddc0dbp-84,HPI=0x1.921fb54442d18p+0;
static void poly(int v,double xr,uint32_t quad,uint32_t sidx,float*sn,float*cs){
    const double C0=0x1p0,C1=-0x1.ffffffd0c621cp-2,C2=0x1.55553e1068f19p-5,C3=-0x1.6c087e89a359dp-10,C4=0x1.99343027bf8c3p-16;
    const double S1=-0x1.555545995a603p-3,S2=0x1.1107605230bc4p-7,S3=-0x1.994eb3774cf24p-13;
    double x2=xr*xr,x3=x2*xr,x4=x2*x2;
    double cc2=mad(v,x2,C4,C3),s1=mad(v,x2,S3,S2),cc1=mad(v,x2,C1,C0);
    double x5=x3*x2,x6=x4*x2;
    double s=mad(v,x3,S1,xr),c=mad(v,x4,C2,cc1);
    uint32_t fs=fb((float)mad(v,x5,s1,s)),fc=fb((float)mad(v,x6,cc2,c));
    if((sidx+1)&2) fs^=0x80000000u;
    if(sidx&2) fc^=0x80000000u;
    if(quad&1){uint32_t t=fs;fs=fc;fc=t;}
    memcpy(sn,&fs,4);memcpy(cs,&fc,4);
}
static void signs(uint32_t n,float s,float c,float*sn,float*cs){ uint32_t fs=fb(s),fc=fb(c);
    if((n+1)&2) fs^=0x80000000u; if(n&2) fc^=0x80000000u; if(n&1){uint32_t t=fs;fs=fc;fc=t;} memcpy(sn,&fs,4);memcpy(cs,&fc,4); }
// round 5, FMA build: Horner's rule on the remainder in radians (dpx_sincos.h, sincos_horner)
static void horner(double x,uint32_t n,float*sn,float*cs){
    const double C0=0x1p0,C1=-0x1.ffffffd0c621cp-2,C2=0x1.55553e1068f19p-5,C3=-0x1.6c087e89a359dp-10,C4=0x1.99343027bf8c3p-16;
    const double S1=-0x1.555545995a603p-3,S2=0x1.1107605230bc4p-7,S3=-0x1.994eb3774cf24p-13;
    double x2=x*x, u=__builtin_fma(x2,__builtin_fma(x2,S3,S2),S1); double s=__builtin_fma(x,x2*u,x);
    double h=__builtin_fma(x2,__builtin_fma(x2,C4,C3),C2); double c=__builtin_fma(x2,__builtin_fma(x2,h,C1),C0);
    signs(n,(float)s,(float)c,sn,cs); }
// ... and on the remainder in quadrants, pi/2 inside the coefficients (sincos_horner_quadrants)
static void horner_quadrants(double r,uint32_t n,float*sn,float*cs){
    const double SP0=0x1.921fb54442d18p+0,SP1=-0x1.4abbbf2376856p-1,SP2=0x1.466031025d4cdp-4,SP3=-0x1.2dd0472562ec7p-8;
    const double CP1=-0x1.3bd3cc7ec2ba7p+0,CP2=0x1.03c1decc70af8p-2,CP3=-0x1.55c6643b8d8a8p-6,CP4=0x1.d9f7bc1fcaa24p-11;
    double r2=r*r, w=__builtin_fma(r2,__builtin_fma(r2,__builtin_fma(r2,SP3,SP2),SP1),SP0); double s=r*w;
    double h=__builtin_fma(r2,__builtin_fma(r2,CP4,CP3),CP2); double c=__builtin_fma(r2,__builtin_fma(r2,h,CP1),1.0);
    signs(n,(float)s,(float)c,sn,cs); }
// large: signed x, magic rounding; FMA build: two-term 2/pi, remainder in quadrants; SSE2 build: three terms, * pi/2, glibc's order
static void quick_large(float y,int v,float*sn,float*cs){
    double x=y; double pm=__builtin_fma(x,C,MAGIC); uint32_t n=(uint32_t)db(pm); double nd=pm-MAGIC;
    double r=__builtin_fma(x,c1,-nd); r=__builtin_fma(x,c2,r);
    if(v){ horner_quadrants(r,n,sn,cs); return; }
    r=__builtin_fma(x,c3,r); double xr=r*HPI;
    poly(v,xr,n,n,sn,cs);
}
// plain: signed x, magic rounding, glibc's own xr = x - n*hpi; returns |xr| for the test
static double quick_plain(float y,int v,float*sn,float*cs){
    double x=y; double pm=__builtin_fma(x,C,MAGIC); uint32_t n=(uint32_t)db(pm); double nd=pm-MAGIC;
    double xr = v? __builtin_fma(-nd,HPI,x) : x-nd*HPI;
    if(v) horner(xr,n,sn,cs); else poly(v,xr,n,n,sn,cs);
    return fabs(xr);
}
// mixed: shared quadrant rounding, both remainders, one selected per lane, Horner in radians; tiny lanes sin = y, cos = 1
static void mixed(float y,float*sn,float*cs){
    double x=y; uint32_t ax=fb(y)&0x7fffffffu; double pm=__builtin_fma(x,C,MAGIC); uint32_t n=(uint32_t)db(pm); double nd=pm-MAGIC;
    double xs=__builtin_fma(-nd,HPI,x); double xl=__builtin_fma(x,c2,__builtin_fma(x,c1,-nd))*HPI;
    horner(ax<0x42f00000u?xs:xl,n,sn,cs);
    if(ax<0x39800000u){ *sn=y; *cs=1.0f; } }
typedef struct { uint32_t lo,hi; int v,mode; uint64_t n,mism; double max_ok_xr, min_bad_xr; uint32_t bads[16]; } job_t;
static void *work(void*a){ job_t*j=a; j->min_bad_xr=10;
  for(uint64_t u=j->lo;u<j->hi;++u){ for(int sg=0;sg<2;++sg){ uint32_t b=(uint32_t)u|(sg?0x80000000u:0); float y;memcpy(&y,&b,4);
    float es,ec; orc_sincosf_glibc235(y,&es,&ec,j->v); float qs,qc; double axr=0;
    if(j->mode==0) quick_large(y,j->v,&qs,&qc); else if(j->mode==2) mixed(y,&qs,&qc); else axr=quick_plain(y,j->v,&qs,&qc);
    int bad = fb(qs)!=fb(es)||fb(qc)!=fb(ec); j->n++;
    if(bad){ if(j->mism<16) j->bads[j->mism]=b; j->mism++; if(axr<j->min_bad_xr)j->min_bad_xr=axr; } else if(axr>j->max_ok_xr) j->max_ok_xr=axr;
  }} return 0; }
int main(int argc,char**argv){ uint32_t lo=0x42f00000u,hi=0x4e000000u; int v=1,nt=8,mode=0;
  for(int i=1;i<argc;++i){ if(!strcmp(argv[i],"--v"))v=atoi(argv[++i]); else if(!strcmp(argv[i],"--lo"))lo=strtoul(argv[++i],0,0); else if(!strcmp(argv[i],"--hi"))hi=strtoul(argv[++i],0,0); else if(!strcmp(argv[i],"--plain"))mode=1; else if(!strcmp(argv[i],"--mixed")){mode=2;lo=0;hi=0x4e000000u;v=1;} }
  pthread_t th[64]; job_t jb[64]; memset(jb,0,sizeof jb); uint64_t span=hi-lo;
  for(int t=0;t<nt;++t){ jb[t].lo=lo+span*t/nt; jb[t].hi=lo+span*(t+1)/nt; jb[t].v=v; jb[t].mode=mode; pthread_create(&th[t],0,work,&jb[t]); }
  uint64_t n=0,m=0; double mx=0,mn=10;
  for(int t=0;t<nt;++t){ pthread_join(th[t],0); n+=jb[t].n; m+=jb[t].mism; if(jb[t].max_ok_xr>mx)mx=jb[t].max_ok_xr; if(jb[t].min_bad_xr<mn)mn=jb[t].min_bad_xr; for(uint64_t k=0;k<jb[t].mism&&k<16;++k) printf("  bad %08x\n",jb[t].bads[k]); }
  printf("mode=%s v=%d range %08x..%08x n=%llu mismatches=%llu  max|xr| among ok=%.17g (pi/4=%.17g) min|xr| among bad=%.17g\n",mode==2?"mixed":mode?"plain":"large",v,lo,hi,(unsigned long long)n,(unsigned long long)m,mx,M_PI/4,mn);
  return m!=0; }
