"""CPU: the host-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (the reference
has latent UB exactly here: unchecked bone / vertex indices in Deform and the morph loops, SURVEY.md
section 5).  GPU sanitizers are unavailable on the pool, so this covers the host "model compile" and
the oracle restatement."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def test_plan_builder_under_asan_ubsan(tmp_path):
    exe = tmp_path / "plan_san"
    cmd = ["g++", "-std=c++17"] + SAN + [os.path.join(ROOT, "tests", "plan_sanitizer_driver.cpp"),
                                         os.path.join(ROOT, "simple_mmd_renderer_amd", "csrc", "plan.cpp"),
                                         os.path.join(ROOT, "simple_mmd_renderer_amd", "csrc", "pmx.cpp"),
                                         os.path.join(ROOT, "simple_mmd_renderer_amd", "csrc", "pmd.cpp"),
                                         os.path.join(ROOT, "simple_mmd_renderer_amd", "csrc", "vmd.cpp"),
                                         os.path.join(ROOT, "simple_mmd_renderer_amd", "csrc", "rig.cpp"),
                                         os.path.join(ROOT, "simple_mmd_renderer_amd", "csrc", "error.cpp"),
                                         "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "built=" in r.stdout and "pmx fuzz:" in r.stdout and "vmd fuzz:" in r.stdout and "ERROR" not in r.stderr


def test_oracle_restatement_under_asan_ubsan(tmp_path):
    """The checker itself: mmdx_oracle.c driven over a golden fixture through a tiny C main."""
    drv = tmp_path / "drv.c"
    drv.write_text(r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
void mmdx_oracle_morph(uint32_t, uint32_t, const int32_t*, const uint32_t*, const uint32_t*, const float*, const float*, float*);
void mmdx_oracle_skin(uint32_t, const float*, const float*, const float*, const int32_t*, const int64_t*, const float*, const float*, float*, float*);
void mmdx_oracle_normalize(uint32_t, int32_t*, int64_t*, float*, const int64_t*);
void mmdx_oracle_repack32(uint32_t, const float*, const float*, const float*, float, float*);
void mmdx_oracle_bone_pose(uint32_t, const uint32_t*, const float*, const float*, const int8_t*, uint32_t, float*);
int mmdx_oracle_bone_solve(uint32_t, const float*, const int64_t*, const int32_t*, const uint16_t*, const float*, float*, void*);
int main(void) {
  enum { NV = 1000, NB = 20, NM = 4, K = 50 };
  float *pos = malloc(NV*12), *nrm = malloc(NV*12), *uv = malloc(NV*8), *w = malloc(NV*16), *pal = malloc(NB*64);
  float *vimg = malloc(NV*12), *op = malloc(NV*12), *on = malloc(NV*12), *v32 = malloc(NV*32), *mval = malloc(NM*K*12);
  int32_t *type = malloc(NV*4), mtype[NM]; int64_t *ids = malloc(NV*32), parent[NB];
  uint32_t moff[NM+1], *midx = malloc(NM*K*4); float rates[NM] = {0.5f, 0.f, 1.f, 2e-8f};
  unsigned s = 7;
  for (int i = 0; i < NV*3; ++i) { s = s*1664525u+1013904223u; pos[i] = (float)(s>>8)/16777216.f; nrm[i] = pos[i]-0.5f; }
  for (int i = 0; i < NV*2; ++i) uv[i] = 0.5f;
  for (int i = 0; i < NV; ++i) { s = s*1664525u+1013904223u; type[i] = (s>>20)%4; for (int k=0;k<4;++k){ s=s*1664525u+1013904223u; ids[4*i+k]=(s>>12)%NB; w[4*i+k]=(float)((s>>4)&255)/255.f; } }
  for (int b = 0; b < NB; ++b) { parent[b] = b ? (b-1)/2 : -1; for (int k=0;k<16;++k) pal[16*b+k] = (k%5==0) ? 1.f : 0.01f*(float)(b+k); }
  for (int m = 0; m <= NM; ++m) moff[m] = m*K;
  for (int m = 0; m < NM; ++m) mtype[m] = m == 1 ? 0 : 1;
  for (int e = 0; e < NM*K; ++e) { s = s*1664525u+1013904223u; midx[e] = (e/K==1) ? (uint32_t)(e%2?2:0) : (s>>10)%NV; mval[3*e]=0.25f; mval[3*e+1]=-0.5f; mval[3*e+2]=0.125f; }
  mmdx_oracle_normalize(NV, type, ids, w, parent);
  mmdx_oracle_morph(NV, NM, mtype, moff, midx, mval, rates, vimg);
  mmdx_oracle_skin(NV, pos, nrm, vimg, type, ids, w, pal, op, on);
  mmdx_oracle_repack32(NV, op, on, uv, 0.1f, v32);
  printf("%g %g\n", op[0], v32[NV*8-1]);
  free(pos); free(nrm); free(uv); free(w); free(pal); free(vimg); free(op); free(on); free(v32); free(mval); free(type); free(ids); free(midx);
  return 0;
}
''')
    exe = tmp_path / "oracle_san"
    cmd = ["gcc", "-std=gnu11"] + SAN + [str(drv), os.path.join(ROOT, "oracle", "mmdx_oracle.c"), "-o", str(exe), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
