#include <immintrin.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static void rsq8(const double* x, double* o){ __m512d v=_mm512_loadu_pd(x); _mm512_storeu_pd(o,_mm512_rsqrt14_pd(v)); }
static double U(uint64_t b){ double d; memcpy(&d,&b,8); return d; }
static uint64_t B(double d){ uint64_t b; memcpy(&b,&d,8); return b; }
int main(){
  // dump table: parity p (exponent 1022+p i.e. x in [0.5,1) or [1,2)), idx 15 bits -> result bits
  for (int p=0;p<2;p++) for (uint32_t i=0;i<32768;i+=8){ double x[8],o[8]; for(int j=0;j<8;j++) x[j]=U(((uint64_t)(1022+p)<<52)|((uint64_t)(i+j)<<37)); rsq8(x,o); for(int j=0;j<8;j++) printf("%d %u %016llx\n",p,i+j,(unsigned long long)B(o[j])); }
  return 0; }
