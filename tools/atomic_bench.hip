// Micro-benchmark: throughput of fp32 atomic adds into an image-sized buffer on gfx950, by scope / privatization.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t pcg(uint32_t x){ x = x*747796405u+2891336453u; x=((x>>((x>>28u)+4u))^x)*277803737u; return (x>>22u)^x; }
__device__ __forceinline__ uint32_t xcc_id(){ uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template<int MODE>
__global__ void k(float* buf, uint32_t npix, uint32_t per_thread, uint32_t copy_stride){
  uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x;
  float* base = buf;
  if (MODE==3 || MODE==4) base = buf + (size_t)xcc_id()*copy_stride;
  for(uint32_t i=0;i<per_thread;i++){
    uint32_t pix = pcg(tid*977u+i*0x9E3779B9u) % npix;
    float* d = base + (size_t)pix*3;
    float v = 1.0f;
    if (MODE==0){ unsafeAtomicAdd(d,v); unsafeAtomicAdd(d+1,v); unsafeAtomicAdd(d+2,v);}      // current
    if (MODE==1){ __hip_atomic_fetch_add(d,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(d+1,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(d+2,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_AGENT);}
    if (MODE==2){ __hip_atomic_fetch_add(d,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(d+1,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(d+2,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP);}
    if (MODE==3){ __hip_atomic_fetch_add(d,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(d+1,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(d+2,v,__ATOMIC_RELAXED,__HIP_MEMORY_SCOPE_WORKGROUP);} // per-XCD copy
    if (MODE==4){ unsafeAtomicAdd(d,v); unsafeAtomicAdd(d+1,v); unsafeAtomicAdd(d+2,v);}      // per-XCD copy, default atomic
    if (MODE==5){ d[0]+=v; d[1]+=v; d[2]+=v; }  // non-atomic RMW (wrong, speed reference)
    if (MODE==6){ float4* q=(float4*)(buf+(size_t)pix*4); // padded pixel, 64-bit packed? use two f32 + one
                  unsafeAtomicAdd((float*)q,v); unsafeAtomicAdd((float*)q+1,v); unsafeAtomicAdd((float*)q+2,v);}
  }
}
template<int MODE> void run(const char* name, float* buf, uint32_t npix, uint32_t stride, size_t bytes){
  const int blocks=2048, threads=256; const uint32_t per=64;
  hipMemset(buf,0,bytes);
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>,dim3(blocks),dim3(threads),0,0,buf,npix,4,stride); hipDeviceSynchronize();
  hipMemset(buf,0,bytes);
  hipEventRecord(a); hipLaunchKernelGGL(k<MODE>,dim3(blocks),dim3(threads),0,0,buf,npix,per,stride); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  double ops = (double)blocks*threads*per*3;
  // verify total
  std::vector<float> h(bytes/4); hipMemcpy(h.data(),buf,bytes,hipMemcpyDeviceToHost); double s=0; for(float x:h) s+=x;
  printf("%-34s npix %8u  %8.3f ms  %7.2f G atomics/s  sum/expected %.6f\n",name,npix,ms,ops/ms/1e6,s/ops);
}
int main(){
  size_t maxbytes = (size_t)8*4096*2048*3*4 + (size_t)4096*2048*16 + 1024; float* buf; hipMalloc(&buf,maxbytes);
  for (uint32_t npix : {512u*256u, 1920u*1080u, 4096u*2048u}){
    uint32_t stride = npix*3; size_t bytes1=(size_t)npix*3*4, bytes8=bytes1*8, bytes4=(size_t)npix*4*4;
    run<0>("unsafeAtomicAdd (current)",buf,npix,stride,bytes1);
    run<1>("fetch_add agent",buf,npix,stride,bytes1);
    run<2>("fetch_add workgroup (1 copy)",buf,npix,stride,bytes1);
    run<3>("per-XCD copy, workgroup scope",buf,npix,stride,bytes8);
    run<4>("per-XCD copy, unsafeAtomicAdd",buf,npix,stride,bytes8);
    run<5>("non-atomic RMW (reference)",buf,npix,stride,bytes1);
    run<6>("padded float4 pixels",buf,npix,stride,bytes4);
  }
  return 0;
}
