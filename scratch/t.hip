#include <hip/hip_runtime.h>
__global__ void k_axpy(const float* x, float* y, float a, int n){int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i]=a*x[i]+y[i];}
__global__ void k_mfma(const float* a, const float* b, float* d){
  typedef float f16v __attribute__((ext_vector_type(16)));
  f16v acc={0};
  int l=threadIdx.x;
  for(int s=0;s<16;s++){ // K=32: A[32][32], B[32][32]
    float av=a[(l&31)*32 + 2*s+(l>>5)];
    float bv=b[(2*s+(l>>5))*32 + (l&31)];
    acc=__builtin_amdgcn_mfma_f32_32x32x2f32(av,bv,acc,0,0,0);
  }
  for(int r=0;r<16;r++){int row=(r&3)+8*(r>>2)+4*(l>>5); d[row*32+(l&31)]=acc[r];}
}
extern "C" int t_axpy(const float* x, float* y, float a, int n, hipStream_t s){
  hipLaunchKernelGGL(k_axpy, dim3((n+255)/256), dim3(256), 0, s, x,y,a,n); return (int)hipGetLastError();}
extern "C" int t_mfma(const float* a,const float* b,float* d, hipStream_t s){
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, s, a,b,d); return (int)hipGetLastError();}
