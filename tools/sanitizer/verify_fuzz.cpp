// Host-only AddressSanitizer / UBSan harness for gl_verify (the one entry point that parses untrusted bytes):
// argv[1] = directory with desc.bin, cap.bin, dig.bin, proof.bin (written by tests/test_verifier.py); verifies the proof,
// then 400 mutations of it (bit flips, truncations, 0xFF path-length bytes, random garbage).  Exit code 0 = the valid
// proof was accepted, every mutation rejected, and the sanitizers stayed silent.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <random>
#include "../../include/plonky2_mi355x.h"
static std::vector<unsigned char> rd(const char* p){ FILE* f=fopen(p,"rb"); std::vector<unsigned char> v; if(!f) return v; int c; while((c=fgetc(f))!=EOF) v.push_back((unsigned char)c); fclose(f); return v; }
int main(int argc, char** argv){
  std::string dir = argc > 1 ? argv[1] : ".";
  auto d=rd((dir+"/desc.bin").c_str()), cap=rd((dir+"/cap.bin").c_str()), dg=rd((dir+"/dig.bin").c_str()), pr=rd((dir+"/proof.bin").c_str());
  if(d.size()!=sizeof(gl_circuit_desc)){ printf("bad desc %zu %zu\n", d.size(), sizeof(gl_circuit_desc)); return 2; }
  int st=gl_verify((const gl_circuit_desc*)d.data(), (const uint64_t*)cap.data(), (const uint64_t*)dg.data(), pr.data(), pr.size());
  printf("valid: %d\n", st);
  std::mt19937_64 rng(1); int rej=0, acc=0;
  for(int it=0; it<400; it++){
    auto b=pr; int kind=it%4;
    if(kind==0){ b[rng()%b.size()]^=1<<(rng()%8); }
    else if(kind==1){ b.resize(rng()%b.size()); }
    else if(kind==2){ size_t k=rng()%b.size(); b[k]=0xFF; b[(k+1)%b.size()]=0xFF; }
    else { for(int j=0;j<64;j++) b[rng()%b.size()]=(unsigned char)rng(); }
    if(b.empty()) b.push_back(0);
    int s=gl_verify((const gl_circuit_desc*)d.data(), (const uint64_t*)cap.data(), (const uint64_t*)dg.data(), b.data(), b.size());
    if(s==0) acc++; else rej++;
  }
  printf("mutations: %d rejected, %d accepted\n", rej, acc);
  return (st == 0 && acc == 0) ? 0 : 1;
}
