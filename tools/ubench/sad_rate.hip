// Micro-benchmark: issue rate + semantics of the byte-SAD VALU instructions on gfx950.
// Used to choose the SAD primitive of the ME kernels (see DESIGN.md "ME kernel").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

template<int MODE> __global__ void __launch_bounds__(256) rate(uint32_t* out, uint32_t seed, int iters){
  uint32_t a0=seed*threadIdx.x+1, a1=a0*7+3, b=a0^0x5a5a5a5a;
  uint64_t r0=((uint64_t)a1<<32)|a0;
  uint64_t q0=0,q1=0,q2=0,q3=0; uint32_t s0=0,s1=0,s2=0,s3=0;
  for(int i=0;i<iters;i++){
    #pragma unroll
    for(int u=0;u<16;u++){
      if(MODE==0){ // v_sad_u8, 4 independent chains
        s0=__builtin_amdgcn_sad_u8(a0,b,s0); s1=__builtin_amdgcn_sad_u8(a1,b,s1);
        s2=__builtin_amdgcn_sad_u8(a0,a1,s2); s3=__builtin_amdgcn_sad_u8(a1,s0,s3);
      } else if(MODE==1){ // v_qsad_pk_u16_u8
        q0=__builtin_amdgcn_qsad_pk_u16_u8(r0,b,q0); q1=__builtin_amdgcn_qsad_pk_u16_u8(r0,a0,q1);
        q2=__builtin_amdgcn_qsad_pk_u16_u8(r0,a1,q2); q3=__builtin_amdgcn_qsad_pk_u16_u8(r0,(uint32_t)q0,q3);
      } else if(MODE==2){ // v_add_u32 baseline
        s0+=a0; s1+=a1; s2+=b; s3+=s0;
      } else if(MODE==3){ // v_min3_u32
        s0=min(min(s0,a0),s1); s1=min(min(s1,a1),s2); s2=min(min(s2,b),s3); s3=min(min(s3,a0),s0);
      } else if(MODE==4){ // v_mqsad_pk_u16_u8
        q0=__builtin_amdgcn_mqsad_pk_u16_u8(r0,b,q0); q1=__builtin_amdgcn_mqsad_pk_u16_u8(r0,a0,q1);
        q2=__builtin_amdgcn_mqsad_pk_u16_u8(r0,a1,q2); q3=__builtin_amdgcn_mqsad_pk_u16_u8(r0,(uint32_t)q0,q3);
      } else if(MODE==5){ // v_lshl_or_b32
        s0=(a0<<16)|s0; s1=(s0<<3)|s1; s2=(s1<<5)|s2; s3=(s2<<7)|s3; a0+=s3;
      }
    }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=s0+s1+s2+s3+(uint32_t)(q0^q1^q2^q3)+(uint32_t)((q0^q1^q2^q3)>>32);
}
__global__ void sem(const uint64_t* r,const uint32_t* s,const uint64_t* acc,uint64_t* o){
  int i=threadIdx.x; o[i]=__builtin_amdgcn_qsad_pk_u16_u8(r[i],s[i],acc[i]);
}
template<int MODE> double run(const char* name,int ops_per_iter,uint32_t* d){
  int iters=4096, blocks=256*8;
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate<MODE><<<blocks,256>>>(d,3,16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); rate<MODE><<<blocks,256>>>(d,3,iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  double winst=(double)blocks*4*iters*16*ops_per_iter; // wave-instructions
  double rate_g=winst/(ms*1e-3)/1e9;
  printf("%-22s %8.3f ms  %8.1f G wave-instr/s  (%.2f cyc/wave-instr/SIMD @2.4GHz,1024 SIMDs)\n",name,ms,rate_g,1024*2.4/rate_g);
  return rate_g;
}
int main(){
  uint32_t* d; CK(hipMalloc(&d,256*8*256*4));
  run<2>("v_add_u32",4,d); run<0>("v_sad_u8",4,d); run<1>("v_qsad_pk_u16_u8",4,d); run<4>("v_mqsad_pk_u16_u8",4,d);
  run<3>("v_min3_u32",4,d); run<5>("v_lshl_or_b32",5,d);
  // semantics check of qsad vs scalar model
  uint64_t hr[64],hacc[64],ho[64]; uint32_t hs[64]; srand(1);
  for(int i=0;i<64;i++){hr[i]=((uint64_t)rand()<<33)^((uint64_t)rand()<<11)^rand(); hs[i]=((uint32_t)rand()<<16)^rand(); hacc[i]=((uint64_t)(rand()&0x3fff)<<48)|((uint64_t)(rand()&0x3fff)<<32)|((uint64_t)(rand()&0x3fff)<<16)|(rand()&0x3fff);}
  hs[0]=0; hs[1]=0x00ff00ff; hr[2]=0;
  uint64_t *dr,*dacc,*dout; uint32_t* ds; CK(hipMalloc(&dr,512));CK(hipMalloc(&dacc,512));CK(hipMalloc(&dout,512));CK(hipMalloc(&ds,256));
  CK(hipMemcpy(dr,hr,512,hipMemcpyHostToDevice));CK(hipMemcpy(dacc,hacc,512,hipMemcpyHostToDevice));CK(hipMemcpy(ds,hs,256,hipMemcpyHostToDevice));
  sem<<<1,64>>>(dr,ds,dacc,dout); CK(hipMemcpy(ho,dout,512,hipMemcpyDeviceToHost));
  int bad=0;
  for(int i=0;i<64;i++){ uint64_t exp=0; for(int k=0;k<4;k++){ uint32_t sad=0; for(int j=0;j<4;j++){int rb=(hr[i]>>(8*(k+j)))&0xff; int sb=(hs[i]>>(8*j))&0xff; sad+=abs(rb-sb);} uint16_t v=(uint16_t)(((hacc[i]>>(16*k))&0xffff)+sad); exp|=(uint64_t)v<<(16*k);} if(exp!=ho[i]){bad++; if(bad<4)printf("qsad mismatch lane %d: got %016llx exp %016llx\n",i,(unsigned long long)ho[i],(unsigned long long)exp);} }
  printf("qsad_pk_u16_u8 semantics: %s (%d mismatches)\n",bad?"DIFFERENT":"MATCH model D.u16[k]=acc[k]+sum_j|ref.b[k+j]-src.b[j]|",bad);
  return 0;
}
