#!/usr/bin/env python3
"""Diagnostic build of the library with in-kernel phase stamps of series_res (never shipped: the shipping sources carry no
stamp code).  Copies povar_amd/csrc to build/variants/res_stamps/povar_amd/csrc/, inserts s_memtime stamps behind the phase comments of
povar_kernels_res.hpp (wavefront 0 and the last wavefront of every workgroup, at term POVAR_RES_STAMP_TERM, default 5) and a
dump of the stamp buffer into povar_synchronize, and builds build/libpovar_hip_res_stamps.so.
Use: POVAR_LIB=build/libpovar_hip_res_stamps.so POVAR_RES_STAMPS_OUT=file python3 tools/res_stamps_report.py shape [world]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "povar_amd", "csrc")
TOP = os.path.join(ROOT, "build", "variants", "res_stamps")
DST = os.path.join(TOP, "povar_amd", "csrc")

# phase comment (a unique substring of a line of the kernel) -> stamp number, inserted BEFORE that line
MARKS = [
    ("    // ---- the norms of term i - 1", 1),
    ("    // ---- hand-over 2: z of the workgroup's cameras into the region", 2),
    ("    __syncthreads();  // B1", 3),      # z gathered
    ("    // ---- forward: u_l += P3^T", 4),  # behind B1
    ("    __syncthreads();  // B2", 5),      # forward done
    ("    __syncthreads();  // B3", 6),      # g = G u done (behind B2)
    ("    __syncthreads();  // B4", 7),      # backward done (behind B3)
    ("    __syncthreads();  // B5", 8),      # records written (behind B4)
    ("    __syncthreads();  // B6", 9),      # owner's records gathered (behind B5)
    ("    __syncthreads();  // B7", 10),     # z published (behind B6)
]


def main():
    shutil.rmtree(TOP, ignore_errors=True)
    shutil.copytree(SRC, DST)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(TOP, "include"))
    p = os.path.join(DST, "povar_kernels_res.hpp")
    s = open(p).read()
    s = s.replace("  unsigned spin_limit;\n};", "  unsigned spin_limit;\n  unsigned long long* stamps;  // [W][2][16]\n  int stamp_term;\n};", 1)
    macro = ('#define RES_STAMP(n) do { if (k.stamps && lane == 0 && (wave == 0 || wave == NW - 1) && i == k.stamp_term) '
             '{ k.stamps[((size_t)g * 2 + (wave != 0)) * 16 + (n)] = __builtin_amdgcn_s_memtime(); k.stamps[(size_t)RES_MAX_WG * 32 + ((size_t)g * 2 + (wave != 0)) * 16 + (n)] = __builtin_amdgcn_s_memrealtime(); } } while (0)\n')
    s = s.replace("// All threads of the workgroup: granule pair i", macro + "// All threads of the workgroup: granule pair i", 1)
    for mark, n in MARKS:
        assert s.count(mark) == 1, (mark, s.count(mark))
        s = s.replace(mark, f"    RES_STAMP({n});\n" + mark, 1)
    # stamp 0: kernel entry, stamp 11: end of the stamped term's loop body, 12: kernel exit
    s = s.replace("  const ResBufs B = res_bufs(k);\n", "  const ResBufs B = res_bufs(k);\n  { const int i = k.stamp_term; RES_STAMP(0); }\n", 1)
    s = s.replace("  // ---------------- epilogue: sum and last term of the owned cameras, status\n", "  { const int i = k.stamp_term; RES_STAMP(12); }\n  // ---------------- epilogue: sum and last term of the owned cameras, status\n", 1)
    open(p, "w").write(s)
    h = os.path.join(DST, "povar_series.hip")  # (res_params lives with the term loop)
    s = open(h).read()
    s = s.replace("  k.spin_limit = c->res_spin_limit;\n", '''  k.spin_limit = c->res_spin_limit;
  {
    static unsigned long long* g_stamps = nullptr;
    if (!g_stamps) { (void)hipMalloc((void**)&g_stamps, sizeof(unsigned long long) * RES_MAX_WG * 2 * 16 * 2); (void)hipMemset(g_stamps, 0, sizeof(unsigned long long) * RES_MAX_WG * 2 * 16 * 2); }
    k.stamps = g_stamps;
    k.stamp_term = std::getenv("POVAR_RES_STAMP_TERM") ? std::atoi(std::getenv("POVAR_RES_STAMP_TERM")) : 5;
    if (const char* f = std::getenv("POVAR_RES_STAMPS_OUT")) {
      static std::string path; path = f;
      static povar_ctx* cc; cc = const_cast<povar_ctx*>(c);
      static bool reg = false;
      if (!reg) { reg = true; std::atexit([]() {
        std::vector<unsigned long long> hbuf((size_t)RES_MAX_WG * 2 * 16 * 2);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hbuf.data(), g_stamps, hbuf.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* o = std::fopen(path.c_str(), "w")) {
          for (int g = 0; g < 2 * RES_MAX_WG; ++g) for (int w = 0; w < 2; ++w) { for (int n = 0; n < 16; ++n) std::fprintf(o, "%llu ", hbuf[((size_t)g * 2 + w) * 16 + n]); std::fprintf(o, "\\n"); }
          std::fclose(o);
        } }); }
    }
  }
''', 1)
    open(h, "w").write(s)
    out = os.path.join(ROOT, "build", "libpovar_hip_res_stamps.so")
    subprocess.check_call(["make", "-s", "-C", DST, f"OUT={out}", f"OBJ_DIR={os.path.join(TOP, 'obj')}", out])  # (the copy's own Makefile: five units)
    print(out)


if __name__ == "__main__":
    sys.exit(main())
