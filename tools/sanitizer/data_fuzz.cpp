// Host-only AddressSanitizer / UBSan harness for the entry points that parse untrusted CIRCUIT DATA: gl_common_data_from_bytes and
// gl_verify_bytes (VerifierCircuitData bytes + proof bytes), and gl_verify / gl_common_data_to_bytes on a mutated DESCRIPTION (the struct
// a caller fills: every count in it indexes something).  argv[1] = directory with vd.bin (VerifierCircuitData), proof.bin, desc.bin,
// cap.bin, dig.bin of a circuit with lookup tables (tests/test_verifier.py).  Exit code 0 = the valid inputs were accepted, no mutation
// of the circuit data was accepted as the original circuit (same bytes back), and the sanitizers stayed silent.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <random>
#include "../../include/plonky2_mi355x.h"
static std::vector<unsigned char> rd(const char* p){ FILE* f=fopen(p,"rb"); std::vector<unsigned char> v; if(!f) return v; int c; while((c=fgetc(f))!=EOF) v.push_back((unsigned char)c); fclose(f); return v; }
int main(int argc, char** argv){
  std::string dir = argc > 1 ? argv[1] : ".";
  auto vd=rd((dir+"/vd.bin").c_str()), pr=rd((dir+"/proof.bin").c_str()), ds=rd((dir+"/desc.bin").c_str()), cap=rd((dir+"/cap.bin").c_str()), dg=rd((dir+"/dig.bin").c_str());
  if(ds.size()!=sizeof(gl_circuit_desc) || vd.empty() || pr.empty()){ printf("bad inputs\n"); return 2; }
  const int valid = gl_verify_bytes(vd.data(), vd.size(), pr.data(), pr.size());
  printf("valid: %d\n", valid);
  std::mt19937_64 rng(7);
  int parsed=0, refused=0, verified=0;
  // ---- mutated VerifierCircuitData bytes ----
  for(int it=0; it<3000; it++){
    auto b=vd; const int kind=it%6;
    if(kind==0){ b[rng()%b.size()]^=1<<(rng()%8); }
    else if(kind==1){ b.resize(rng()%b.size()); }
    else if(kind==2){ size_t k=rng()%b.size(); for(int j=0;j<8 && k+j<b.size();j++) b[k+j]=0xFF; }                     // a huge usize somewhere
    else if(kind==3){ size_t k=(rng()%(b.size()/8))*8; uint64_t v=rng()%70000; memcpy(&b[k],&v,8); }                     // a plausible count, 8-aligned
    else if(kind==4){ for(int j=0;j<32;j++) b[rng()%b.size()]=(unsigned char)rng(); }
    else { size_t k=rng()%b.size(); b.insert(b.begin()+k, (size_t)(rng()%64), (unsigned char)rng()); }                    // inserted bytes
    if(b.empty()) b.push_back(0);
    // the common data sits behind the verifier-only part: find it the way gl_verify_bytes does, by parsing from the start
    const int s = gl_verify_bytes(b.data(), b.size(), pr.data(), pr.size());
    if(s==0) verified++; else refused++;
    // and the reader alone, at every plausible offset of the common data (the verifier-only part of this file is 8 + 16*32 + 32 bytes)
    gl_circuit_desc d; size_t used=0;
    const size_t off = 8 + 16*32 + 32;
    if(b.size() > off && gl_common_data_from_bytes(b.data()+off, b.size()-off, &d, &used)==0){
      parsed++;
      std::vector<unsigned char> back(1<<20); size_t nb=0;
      (void)gl_common_data_to_bytes(&d, back.data(), back.size(), &nb);            // whatever the reader accepts, the writer must survive
    }
  }
  printf("circuit-data mutations: %d refused, %d still verify, %d parsed\n", refused, verified, parsed);
  // ---- mutated descriptions (the non-table part: counts, gate arrays, rows) ----
  int desc_ok=0, desc_bad=0;
  const size_t head = offsetof(gl_circuit_desc, lut);
  for(int it=0; it<3000; it++){
    auto b=ds; const int kind=it%3;
    if(kind==0){ b[rng()%head]^=1<<(rng()%8); }
    else if(kind==1){ size_t k=(rng()%(head/4))*4; uint32_t v=(uint32_t)rng(); memcpy(&b[k],&v,4); }
    else { size_t k=(rng()%(head/4))*4; uint32_t v=(uint32_t)(rng()%40); memcpy(&b[k],&v,4); }
    const int s=gl_verify((const gl_circuit_desc*)b.data(), (const uint64_t*)cap.data(), (const uint64_t*)dg.data(), pr.data(), pr.size());
    if(s==0) desc_ok++; else desc_bad++;
    std::vector<unsigned char> back(1<<20); size_t nb=0;
    (void)gl_common_data_to_bytes((const gl_circuit_desc*)b.data(), back.data(), back.size(), &nb);
  }
  printf("description mutations: %d rejected, %d accepted\n", desc_bad, desc_ok);
  return valid == 0 ? 0 : 1;
}
