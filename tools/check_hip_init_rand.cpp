#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
int main(){
  srand(10); int a=rand(), b=rand(); printf("ref: %d %d\n",a,b);
  srand(10);
  int n=0; hipGetDeviceCount(&n);
  printf("after hipGetDeviceCount: %d\n", rand());
  srand(10);
  hipStream_t s; hipStreamCreate(&s);
  printf("after hipStreamCreate: %d\n", rand());
  srand(10);
  void* p; hipMalloc(&p, 1<<20);
  printf("after hipMalloc: %d\n", rand());
  srand(10);
  hipMemsetAsync(p,0,1<<20,s); hipStreamSynchronize(s);
  printf("after memset+sync: %d\n", rand());
  srand(10);
  hipStream_t s2; hipStreamCreate(&s2); hipStreamSynchronize(s2);
  printf("after 2nd hipStreamCreate: %d\n", rand());
  return 0;
}
