#!/usr/bin/env python3
"""Regenerates tools/variants/ck_stamps.patch from the CURRENT shipping sources: in-kernel s_memtime stamps of e0_ck's phases
(tools/ck_stamps.py) and the timing-only experiment branches (-DPOVAR_CK_EXP_*; results wrong by construction).  The edits are
anchored on source text, so a change of the kernel that moves an anchor fails loudly here instead of leaving a stale patch.
    python tools/variants/make_ck_stamps_patch.py && tools/variants/build_variant.sh ck_stamps stamps"""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sub(s, old, new, count=1):
    assert s.count(old) >= 1, "anchor not found:\n" + old
    return s.replace(old, new, count)


def main():
    top = tempfile.mkdtemp(prefix="ckst")
    for ab in "ab":
        os.makedirs(os.path.join(top, ab, "povar_amd"))
        shutil.copytree(os.path.join(ROOT, "povar_amd", "csrc"), os.path.join(top, ab, "povar_amd", "csrc"),
                        ignore=shutil.ignore_patterns("host", "*.so", "*.o"))  # (Makefile and povar.map travel: the variant is built with them)
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(top, ab, "include"))
    b = os.path.join(top, "b", "povar_amd", "csrc")

    # ---- the stamp buffer (povar_ctx.hpp), its release (povar_create.hip), the kernel argument and povar_debug_ck_stamps (povar_series.hip)
    p = os.path.join(b, "povar_ctx.hpp")
    s = open(p).read()
    s = sub(s, "    ckh, pl_ckh;     // step 2 (e0_ck_h): a second instance (64 bytes of LDS per landmark slot: more batches, other chunks)\n",
            "    ckh, pl_ckh;     // step 2 (e0_ck_h): a second instance (64 bytes of LDS per landmark slot: more batches, other chunks)\n"
            "  DevBuf<unsigned long long> ck_stamps;  // diagnostic build: CkP::stamps\n")
    open(p, "w").write(s)
    p = os.path.join(b, "povar_create.hip")
    s = open(p).read()
    s = sub(s, "c->ckh.release(); c->pl_ckh.release(); c->ck_zero_range.release();", "c->ckh.release(); c->pl_ckh.release(); c->ck_zero_range.release(); c->ck_stamps.release();")
    open(p, "w").write(s)
    p = os.path.join(b, "povar_series.hip")
    s = open(p).read()
    s = sub(s, "D.packed ? 1 : 0, D.cold_q ? D.cpos.p : nullptr, c->q4c.p};", "D.packed ? 1 : 0, D.cold_q ? D.cpos.p : nullptr, c->q4c.p, c->ck_stamps.p};")
    i = s.index("int povar_debug_ck_stamps(povar_ctx* c, uint64_t* out, int64_t n) {")
    j = s.index("\n}\n", i) + 3
    s = s[:i] + """int povar_debug_ck_stamps(povar_ctx* c, uint64_t* out, int64_t n) {
  if (int rc = check_ctx(c)) return rc;
  const size_t want = (size_t)c->e0c_grid * 16 * CK_N_STAMPS;
  if (!c->ck_stamps.p) {  // first call: allocate; the stamps of the launches from now on are returned by the next call
    HIP_TRY(c->ck_stamps.alloc(want, &c->bytes));
    HIP_TRY(hipMemset(c->ck_stamps.p, 0, want * sizeof(unsigned long long)));
    if (c->series_graph) { (void)hipGraphExecDestroy(c->series_graph); c->series_graph = nullptr; }
    return 0;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(out, c->ck_stamps.p, std::min<size_t>((size_t)n, want) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return (int)std::min<size_t>((size_t)n, want);
}
""" + s[j:]
    open(p, "w").write(s)

    # ---- povar_kernels.hpp: the per-camera kernel without its partial-record loads
    p = os.path.join(b, "povar_kernels.hpp")
    s = open(p).read()
    i = s.index("__global__ __launch_bounds__(NT) void cam_cold_sum_binv(Dp d, int want_norms) {")
    old = "  if (d.part_range) {  // e0_lpl: the camera's partial records are one contiguous run\n    for (int w = rr.x + t; w < rr.y; w += NT) {"
    k = s.index(old, i)
    e = s.index("  block_sum_dpp<12, NT>(acc, sh);  // every thread now holds the 12 sums", k)
    s = s[:k] + "#if !defined(POVAR_CK_EXP_NOPART) && !defined(POVAR_CK_EXP_NOPART_LOADS)\n" + s[k:e] + \
        "#else\n  acc[0] = (double)(rr.y - rr.x + r);\n#endif\n" + s[e:]
    open(p, "w").write(s)

    # ---- povar_kernels_ck.hpp
    p = os.path.join(b, "povar_kernels_ck.hpp")
    s = open(p).read()
    s = sub(s, "  double4* q4c;            // ... such a lane stores q of every observation there (the per-camera kernel forms h~ (x) q), no record\n};",
            "  double4* q4c;            // ... such a lane stores q of every observation there (the per-camera kernel forms h~ (x) q), no record\n"
            "  unsigned long long* stamps;  // diagnostic build: [grid][16][CK_N_STAMPS] s_memtime stamps, else nullptr\n};\n"
            "constexpr int CK_N_STAMPS = 40;\n"
            "#ifndef POVAR_CK_NO_STAMPS  // (the timing-only experiment builds are compiled with -DPOVAR_CK_NO_STAMPS)\n"
            "#define CK_STAMP(i)                                                                                   \\\n"
            "  do {                                                                                                \\\n"
            "    if (k.stamps && lane0 == 0 && (i) < CK_N_STAMPS)                                                  \\\n"
            "      k.stamps[((size_t)blockIdx.x * 16 + wave_all) * CK_N_STAMPS + (i)] = __builtin_amdgcn_s_memtime(); \\\n"
            "  } while (0)\n#else\n#define CK_STAMP(i)\n#endif")
    s = sub(s, "  __hip_atomic_fetch_add(lu + s, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (s = 3 x slot)\n"
               "  __hip_atomic_fetch_add(lu + s + 1, red[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
               "  __hip_atomic_fetch_add(lu + s + 2, red[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n",
            "#ifdef POVAR_CK_EXP_NOATOMIC  // timing-only: plain stores instead of the three LDS atomics\n"
            "  lu[s] = red[0]; lu[s + 1] = red[1]; lu[s + 2] = red[2];\n#else\n"
            "  __hip_atomic_fetch_add(lu + s, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (s = 3 x slot)\n"
            "  __hip_atomic_fetch_add(lu + s + 1, red[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n"
            "  __hip_atomic_fetch_add(lu + s + 2, red[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);\n#endif\n")
    # rows: the way forward / back re-reading ONE row per tile
    s = sub(s, "  st.load(k, row0, li0, j + D, h, lane, i);\n",
            "#ifdef POVAR_CK_EXP_NOFWDROWS  // timing-only: the way forward re-reads ONE row of the tile\n  st.load(k, row0, li0, 0, h, lane, i);\n#else\n"
            "  st.load(k, row0, li0, j + D, h, lane, i);\n#endif\n")
    s = sub(s, "  st.load(k, row0, li0, j - D, h, lane, i);\n",
            "#ifdef POVAR_CK_EXP_NOBWDROWS  // timing-only: the way back re-reads ONE row of the tile\n  st.load(k, row0, li0, 0, h, lane, i);\n#else\n"
            "  st.load(k, row0, li0, j - D, h, lane, i);\n#endif\n")
    # the camera gathers: every lane the SAME record
    s = sub(s, "__device__ inline void ck_load_z_img(const Dp& d, int rank, double* zz) {\n",
            "#ifdef POVAR_CK_EXP_UNIGATHER  // timing-only: every lane gathers the SAME record (one cache line per load instead of 64)\n"
            "#define CK_GATHER_RANK(r) 0\n#else\n#define CK_GATHER_RANK(r) (r)\n#endif\n"
            "__device__ inline void ck_load_z_img(const Dp& d, int rank, double* zz) {\n  rank = CK_GATHER_RANK(rank);\n")
    s = sub(s, "__device__ inline void ck_load_p3(const Dp& d, int rank, double* P3) {\n",
            "__device__ inline void ck_load_p3(const Dp& d, int rank, double* P3) {\n  rank = CK_GATHER_RANK(rank);\n")
    # partial records
    s = sub(s, "      const __amdgpu_buffer_rsrc_t part_out = ck_part_rsrc(part_ptr);\n      const unsigned o = (unsigned)(~acc_slot) * 96u;\n#pragma unroll\n"
               "      for (int m = 0; m < 6; ++m) ck_store_part(part_out, o + 16u * m, y[2 * m], y[2 * m + 1]);\n",
            "#if !defined(POVAR_CK_EXP_NOPART) && !defined(POVAR_CK_EXP_NOPART_COLD)  // timing-only: no partial record leaves / none of a chunk's own\n"
            "      const __amdgpu_buffer_rsrc_t part_out = ck_part_rsrc(part_ptr);\n      const unsigned o = (unsigned)(~acc_slot) * 96u;\n#pragma unroll\n"
            "      for (int m = 0; m < 6; ++m) ck_store_part(part_out, o + 16u * m, y[2 * m], y[2 * m + 1]);\n"
            "#else\n      if (y[0] == 1.2345e300) part_ptr[0] = y[1];  // (keeps the sums alive)\n#endif\n")
    s = sub(s, "    const int rec = k.slot_rec[cam0 + r];\n    ck_store_part(",
            "    const int rec = k.slot_rec[cam0 + r];\n#if defined(POVAR_CK_EXP_NOPART) || defined(POVAR_CK_EXP_NOPART_FLUSH)\n"
            "    if (acc[r * CK_ACC_STRIDE + m] == 1.2345e300)\n#endif\n    ck_store_part(")
    # stamps
    s = sub(s, "    asm volatile(\"\" : \"+v\"(lane));\n    const int tb0 = bt_of(b, b == grp), tb1 = bt_of(b + 1, b == grp);\n    int q_t = 0;",
            "    asm volatile(\"\" : \"+v\"(lane));\n    CK_STAMP(8 * (b / NG) + 0);\n    const int tb0 = bt_of(b, b == grp), tb1 = bt_of(b + 1, b == grp);\n    int q_t = 0;")
    s = sub(s, "    if (t < tb1) ck_load_z_img(d, rank < 0 ? 0 : rank, zz);\n", "    if (t < tb1) ck_load_z_img(d, rank < 0 ? 0 : rank, zz);\n    CK_STAMP(8 * (b / NG) + 1);\n")
    s = sub(s, "    request_next_fwd();\n    group_barrier();\n", "    request_next_fwd();\n    CK_STAMP(8 * (b / NG) + 2);\n    group_barrier();\n    CK_STAMP(8 * (b / NG) + 3);\n")
    s = sub(s, "      ck_forward_rows<SD, ROBUST, PK>(d, R, st, row0, li0, h, lane, zz, P3, lh, lu, S);\n",
            "      ck_forward_rows<SD, ROBUST, PK>(d, R, st, row0, li0, h, lane, zz, P3, lh, lu, S);\n      if (b < NG) CK_STAMP(20 + 2 * q_t);\n")
    s = sub(s, "    // ---- the way back starts before the barriers in front of it:", "    CK_STAMP(8 * (b / NG) + 4);\n    // ---- the way back starts before the barriers in front of it:")
    s = sub(s, "      st.template start<-1>(R, row0, li0, h, lane);\n    }\n    group_barrier();\n    request_first_meta(b + NG, lane);",
            "      CK_STAMP(24 + 4 * (b / NG));      // G + metadata of the way back requested\n"
            "      st.template start<-1>(R, row0, li0, h, lane);\n    }\n"
            "    CK_STAMP(25 + 4 * (b / NG));        // ... and its first rows\n"
            "    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
            "    CK_STAMP(26 + 4 * (b / NG));        // LDS atomics + scalar loads drained (what the barrier's lgkmcnt(0) waits for)\n"
            "    group_barrier();\n    CK_STAMP(8 * (b / NG) + 5);\n    request_first_meta(b + NG, lane);")
    s = sub(s, "    request_next_bwd();\n    group_barrier();\n", "    request_next_bwd();\n    group_barrier();\n    CK_STAMP(8 * (b / NG) + 6);\n")
    s = sub(s, "    group_barrier();  // the next batch overwrites h~ and u; after the last one: the accumulators are complete\n",
            "    CK_STAMP(8 * (b / NG) + 7);\n    group_barrier();  // the next batch overwrites h~ and u; after the last one: the accumulators are complete\n")
    s = sub(s, "  // ---- accumulators -> this workgroup's partial records (camera-major in part_out)\n  const __amdgpu_buffer_rsrc_t PR",
            "  CK_STAMP(8 * (k.nb / NG));\n  // ---- accumulators -> this workgroup's partial records (camera-major in part_out)\n  const __amdgpu_buffer_rsrc_t PR")
    s = sub(s, "  if (d.p2p_epoch && blockIdx.x == 0 && threadIdx.x == 0) *d.p2p_epoch += 1;  // one tick per term (as e0_lpl)",
            "  CK_STAMP(8 * (k.nb / NG) + 1);\n  if (d.p2p_epoch && blockIdx.x == 0 && threadIdx.x == 0) *d.p2p_epoch += 1;  // one tick per term (as e0_lpl)")
    open(p, "w").write(s)

    out = os.path.join(ROOT, "tools", "variants", "ck_stamps.patch")
    with open(out, "w") as fh:
        subprocess.run(["diff", "-ruN", "a", "b"], cwd=top, stdout=fh)
    shutil.rmtree(top)
    print(out, sum(1 for _ in open(out)), "lines")


if __name__ == "__main__":
    main()
