#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
static uint32_t code2b(uint32_t c){uint32_t x=(c>>1)&3u;return x^(x>>1);}
static int is_acgt(uint32_t c){uint32_t u=c&0xDFu;return (u=='A')|(u=='C')|(u=='G')|(u=='T');}
static uint32_t ref(const uint32_t w[4], uint32_t *bad){uint32_t out=0;*bad=0;for(int i=0;i<4;i++)for(int j=0;j<4;j++){uint32_t c=(w[i]>>(8*j))&0xFFu;int idx=4*i+j;out|=code2b(c)<<(30-2*idx);*bad|=(is_acgt(c)?0u:1u)<<idx;}return out;}
static uint32_t codes4(uint32_t c){uint32_t x=(c>>1)&0x03030303u;x^=(x>>1)&0x01010101u;return (x*0x40100401u)>>24;}
static uint32_t diff4(uint32_t c){uint32_t u=c&0xDFDFDFDFu;uint32_t t=(u>>2)&~(u>>1)&0x01010101u;uint32_t e=0x40404040u|(u&0x06060606u)|(t<<4)|(t^0x01010101u);return u^e;}
static uint32_t bad4(uint32_t d){uint32_t nz=(((d&0x7F7F7F7Fu)+0x7F7F7F7Fu)|d)&0x80808080u;return (((nz>>7)*0x01020408u)>>24)&0xFu;}
static uint32_t swar(const uint32_t w[4], uint32_t *bad){uint32_t out=(codes4(w[0])<<24)|(codes4(w[1])<<16)|(codes4(w[2])<<8)|codes4(w[3]);uint32_t d0=diff4(w[0]),d1=diff4(w[1]),d2=diff4(w[2]),d3=diff4(w[3]);*bad=0;if(d0|d1|d2|d3)*bad=bad4(d0)|(bad4(d1)<<4)|(bad4(d2)<<8)|(bad4(d3)<<12);return out;}
int main(){uint64_t n=0;uint32_t s=12345;
 for(int pos=0;pos<16;pos++)for(int c=0;c<256;c++)for(int rep=0;rep<64;rep++){uint32_t w[4];for(int i=0;i<4;i++){s=s*1664525u+1013904223u;uint32_t r=s; if(rep&1){ /* valid bytes */ const char*L="ACGTacgt";uint32_t v=0;for(int j=0;j<4;j++){s=s*1664525u+1013904223u;v|=(uint32_t)L[(s>>13)&7]<<(8*j);}r=v;} w[i]=r;}
  w[pos>>2]=(w[pos>>2]&~(0xFFu<<(8*(pos&3))))|((uint32_t)c<<(8*(pos&3)));
  uint32_t b1,b2;uint32_t o1=ref(w,&b1),o2=swar(w,&b2);if(o1!=o2||b1!=b2){printf("MISMATCH pos %d c %d: %08x %08x bad %04x %04x\n",pos,c,o1,o2,b1,b2);return 1;}n++;}
 printf("ok %llu cases\n",(unsigned long long)n);return 0;}
