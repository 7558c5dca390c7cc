#!/usr/bin/env python3
"""[developer measurement tool] What a gfx950 SIMD sustains on THIS kernel's instruction mix ("weak" 2).
1. histogram of the VALU opcodes in the encoder kernel's ISA (a -S compile of imcvt_amd/csrc/hevc_hip.hip with the shipped flags; static counts);
2. a generated micro-benchmark: every opcode that makes up >= 0.4 % of them, 8 independent chains, measured with ONE wave per SIMD (what a
   lone wavefront can issue: the latency-bound shapes) and with FOUR waves per SIMD on all 256 compute units (what the SIMD sustains: the
   throughput shapes), timed inside the kernel (s_memtime, first start to last end over a workgroup's wavefronts);
3. the mix-weighted cycles per wave-instruction -> the issue peak bench.py's roofline_issue uses.
usage (GPU box): python tools/valu_mix_probe.py [profiles/r05_valu_mix.json]"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_valu_mix.json")

# opcode (as the disassembler prints it) -> asm template; {n} = this chain's register, {x} = a loop-invariant VGPR, {s} = an SGPR, {m} = an SGPR pair (lane mask)
T = {
    "v_mov_b32_e32": "v_mov_b32 {n}, {x}", "v_add_u32_e32": "v_add_u32 {n}, {n}, {x}", "v_sub_u32_e32": "v_sub_u32 {n}, {n}, {x}", "v_subrev_u32_e32": "v_subrev_u32 {n}, {n}, {x}",
    "v_cndmask_b32_e64": "v_cndmask_b32_e64 {n}, {n}, {x}, {m}", "v_cndmask_b32_e32": "v_cmp_lt_u32 vcc, {n}, {x}\nv_cndmask_b32 {n}, {n}, {x}, vcc",
    "v_lshrrev_b32_e32": "v_lshrrev_b32 {n}, 1, {n}", "v_lshlrev_b32_e32": "v_lshlrev_b32 {n}, 1, {n}", "v_lshlrev_b32_e64": "v_lshlrev_b32_e64 {n}, {s}, {n}", "v_ashrrev_i32_e32": "v_ashrrev_i32 {n}, 1, {n}",
    "v_and_b32_e32": "v_and_b32 {n}, {n}, {x}", "v_or_b32_e32": "v_or_b32 {n}, {n}, {x}", "v_xor_b32_e32": "v_xor_b32 {n}, {n}, {x}",
    "v_add3_u32": "v_add3_u32 {n}, {n}, {x}, {x}", "v_or3_b32": "v_or3_b32 {n}, {n}, {x}, {x}", "v_and_or_b32": "v_and_or_b32 {n}, {n}, {x}, {x}",
    "v_cmp_lt_i32_e64": "v_cmp_lt_i32_e64 s[20:21], {n}, {x}", "v_cmp_lt_i32_e32": "v_cmp_lt_i32 vcc, {n}, {x}", "v_cmp_ne_u32_e64": "v_cmp_ne_u32_e64 s[20:21], {n}, {x}", "v_cmp_ne_u32_e32": "v_cmp_ne_u32 vcc, {n}, {x}",
    "v_cmp_eq_u32_e64": "v_cmp_eq_u32_e64 s[20:21], {n}, {x}", "v_cmp_eq_u32_e32": "v_cmp_eq_u32 vcc, {n}, {x}", "v_cmp_gt_i32_e64": "v_cmp_gt_i32_e64 s[20:21], {n}, {x}", "v_cmp_gt_i32_e32": "v_cmp_gt_i32 vcc, {n}, {x}",
    "v_cmp_gt_u32_e32": "v_cmp_gt_u32 vcc, {n}, {x}", "v_cmp_gt_u32_e64": "v_cmp_gt_u32_e64 s[20:21], {n}, {x}", "v_cmp_lt_u32_e64": "v_cmp_lt_u32_e64 s[20:21], {n}, {x}", "v_cmp_le_i32_e32": "v_cmp_le_i32 vcc, {n}, {x}",
    "v_cmp_eq_u32_sdwa": "v_cmp_eq_u32_sdwa vcc, {n}, {x} src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_gt_u32_sdwa": "v_cmp_gt_u32_sdwa vcc, {n}, {x} src0_sel:BYTE_0 src1_sel:DWORD", "v_cmp_lt_u32_sdwa": "v_cmp_lt_u32_sdwa vcc, {n}, {x} src0_sel:BYTE_0 src1_sel:DWORD",
    "v_bfe_u32": "v_bfe_u32 {n}, {n}, 1, 9", "v_lshl_add_u32": "v_lshl_add_u32 {n}, {n}, 1, {x}", "v_lshl_or_b32": "v_lshl_or_b32 {n}, {n}, 1, {x}",
    "v_readlane_b32": "v_readlane_b32 s20, {n}, 3", "v_readfirstlane_b32": "v_readfirstlane_b32 s20, {n}", "v_writelane_b32": "v_writelane_b32 {n}, {s}, 3",
    "v_min_u32_e32": "v_min_u32 {n}, {n}, {x}", "v_max_u32_e32": "v_max_u32 {n}, {n}, {x}", "v_min_i32_e32": "v_min_i32 {n}, {n}, {x}", "v_max_i32_e32": "v_max_i32 {n}, {n}, {x}", "v_med3_i32": "v_med3_i32 {n}, {n}, {x}, {x}",
    "v_bitop3_b32": "v_bitop3_b32 {n}, {n}, {x}, {x} bitop3:0x48", "v_bitop3_b16": "v_bitop3_b16 {n}, {n}, {x}, {x} bitop3:0x48",
    "v_mul_i32_i24_sdwa": "v_mul_i32_i24_sdwa {n}, {n}, {x} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD", "v_add_u32_sdwa": "v_add_u32_sdwa {n}, {n}, {x} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD",
    "v_sub_u32_sdwa": "v_sub_u32_sdwa {n}, {n}, {x} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD",
    "v_mul_i32_i24_e32": "v_mul_i32_i24 {n}, {n}, {x}", "v_mul_u32_u24_e32": "v_mul_u32_u24 {n}, {n}, {x}", "v_mul_lo_u32": "v_mul_lo_u32 {n}, {n}, {x}", "v_mad_i32_i24": "v_mad_i32_i24 {n}, {n}, {x}, {x}", "v_mad_u32_u24": "v_mad_u32_u24 {n}, {n}, {x}, {x}",
    "v_ffbh_u32_e32": "v_ffbh_u32 {n}, {n}", "v_addc_co_u32_e64": "v_addc_co_u32_e64 {n}, s[20:21], {n}, {x}, {m}", "v_perm_b32": "v_perm_b32 {n}, {n}, {x}, {x}", "v_alignbit_b32": "v_alignbit_b32 {n}, {n}, {x}, 3",
}


def histogram():
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "k.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-mllvm", "-disable-machine-licm", "-S", "--cuda-device-only",
                        os.path.join(ROOT, "imcvt_amd", "csrc", "hevc_hip.hip"), "-o", s], check=True, stderr=subprocess.DEVNULL)
        c = collections.Counter()
        for ln in open(s):
            t = ln.strip().split()
            if t and t[0].startswith("v_") and not t[0].startswith("v_mfma"):
                c[t[0]] += 1
    return c


def gen(ops):
    body = ["#include <hip/hip_runtime.h>", "#include <stdio.h>", "#include <algorithm>", "#include <vector>",
            "#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, \"%s: %s\\n\", #x, hipGetErrorString(e_)); return 1; } } while (0)"]
    for i, op in enumerate(ops):
        per = T[op].count("\n") + 1
        txt = ""
        for rep in range(8):
            for ch in range(8):
                txt += T[op].format(n=f"%{ch}", x="%8", s="%9", m="%10") + "\n"
        txt = txt.replace("\n", "\\n")
        body.append(f"__global__ void k{i}(unsigned long long *out, int iters, unsigned seed) {{\n"
                    "  extern __shared__ unsigned lds[];\n  unsigned r[8]; for (int c = 0; c < 8; c++) r[c] = seed + threadIdx.x + c; unsigned x = seed | 1u, s = seed & 7u; unsigned long long m = 0x3333333333333333ull;\n"
                    "  asm volatile(\"v_cmp_lt_u32 vcc, %0, %1\" : : \"v\"(threadIdx.x & 63u), \"v\"(32u) : \"vcc\");\n  lds[threadIdx.x] = seed; __syncthreads();\n"
                    "  unsigned long long t0 = 0, t1 = 0, tb = 0;\n  for (int pass = 0; pass < 2; pass++) { t0 = __builtin_readcyclecounter(); if (pass == 0) tb = t0; for (int it = 0; it < iters; it++)\n"
                    f"    asm volatile(\"{txt}\" : \"+v\"(r[0]), \"+v\"(r[1]), \"+v\"(r[2]), \"+v\"(r[3]), \"+v\"(r[4]), \"+v\"(r[5]), \"+v\"(r[6]), \"+v\"(r[7]) : \"v\"(x), \"s\"(s), \"s\"(m) : \"vcc\", \"s20\", \"s21\", \"scc\");\n"
                    "    t1 = __builtin_readcyclecounter(); }\n  unsigned a = 0; for (int c = 0; c < 8; c++) a ^= r[c]; if (a == 0x12345u) out[16380] = a;\n"
                    "  if ((threadIdx.x & 63u) == 0) { out[blockIdx.x * 48 + (threadIdx.x >> 6) * 3] = tb; out[blockIdx.x * 48 + (threadIdx.x >> 6) * 3 + 1] = t0; out[blockIdx.x * 48 + (threadIdx.x >> 6) * 3 + 2] = t1; }\n}\n"
                    f"static const int per{i} = {per};")
    body.append("typedef void (*kern_t)(unsigned long long *, int, unsigned);")
    body.append("static kern_t kerns[] = {" + ", ".join(f"k{i}" for i in range(len(ops))) + "};")
    body.append("static const int pers[] = {" + ", ".join(f"per{i}" for i in range(len(ops))) + "};")
    body.append("static const char *names[] = {" + ", ".join(f'"{op}"' for op in ops) + "};")
    body.append(r"""
int main() {
    unsigned long long *d; CHK(hipMalloc(&d, 8 * 16384));
    const int iters = 1000;
    for (size_t i = 0; i < sizeof(kerns) / sizeof(kerns[0]); i++) {
        double res[2];
        for (int cfg = 0; cfg < 2; cfg++) {
            const int blocks = cfg ? 256 : 8, threads = cfg ? 1024 : 256;
            CHK(hipFuncSetAttribute((const void *)kerns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
            CHK(hipMemset(d, 0, 8 * 16384));
            hipLaunchKernelGGL(kerns[i], dim3(blocks), dim3(threads), 100 * 1024, 0, d, iters, 12345u);
            CHK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(blocks * 48); CHK(hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost));
            std::vector<double> v;
            const int nw = threads / 64;
            for (int b = 0; b < blocks; b++) {
                if (cfg == 0) { for (int w = 0; w < nw; w++) v.push_back((double)(h[b * 48 + 3 * w + 2] - h[b * 48 + 3 * w + 1]) / ((double)iters * 64 * pers[i])); }      // a lone wave: its own second pass
                else {                                              // a full SIMD: first start to last end of the workgroup (= compute unit) over BOTH passes (older wavefronts run ahead of younger ones)
                    unsigned long long lo = ~0ull, hi = 0;
                    for (int w = 0; w < nw; w++) { lo = std::min(lo, h[b * 48 + 3 * w]); hi = std::max(hi, h[b * 48 + 3 * w + 2]); }
                    v.push_back((double)(hi - lo) / (2.0 * iters * 64 * pers[i] * (nw / 4)));      // cycles per wave-instruction per SIMD
                }
            }
            std::sort(v.begin(), v.end());
            res[cfg] = v[v.size() / 2];
        }
        printf("%-24s %8.3f %8.3f\n", names[i], res[0], res[1]);
        fflush(stdout);
    }
    return 0;
}
""")
    return "\n".join(body)


hist = histogram()
total = sum(hist.values())
ops = [op for op, n in hist.most_common() if op in T and n / total >= 0.004]
missing = [(op, n) for op, n in hist.most_common() if op not in T and n / total >= 0.004]
with tempfile.TemporaryDirectory() as d:
    src = os.path.join(d, "p.hip"); exe = os.path.join(d, "p")
    open(src, "w").write(gen(ops))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O1", src, "-o", exe], check=True, stderr=subprocess.DEVNULL)
    txt = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
rows = {}
for ln in txt.splitlines():
    f = ln.split()
    if len(f) == 3:
        rows[f[0]] = (float(f[1]), float(f[2]))
# v_cndmask_b32_e32 was measured as a v_cmp + v_cndmask pair (the VOP2 form reads vcc; alone, back to back on a stale vcc, it costs 18 cycles: profiles/r05_valu_sgpr.log)
covered = sum(hist[o] for o in rows)
lone = sum(hist[o] * rows[o][0] for o in rows) / covered
simd = sum(hist[o] * rows[o][1] for o in rows) / covered
out = {"what": "cycles per wave64 VALU instruction: one wave per SIMD on 8 compute units (lone) / four waves per SIMD on all 256 compute units (simd, = what a SIMD sustains), 8 independent chains; "
               "weights = static opcode counts of the encoder kernel's ISA (hipcc -S, shipped flags)",
       "valu_instructions_in_kernel": total, "share_covered": round(covered / total, 4), "not_measured": [(o, n) for o, n in missing],
       "mix_weighted_cycles_lone_wave": round(lone, 3), "mix_weighted_cycles_simd": round(simd, 3),
       "issue_peak_G_wave_inst_per_s": round(256 * 4 * 2.4 / simd, 1), "clock_ghz": 2.4,
       "per_opcode": {o: {"share": round(hist[o] / total, 4), "cycles_lone_wave": rows[o][0], "cycles_simd_4_waves": rows[o][1]} for o in sorted(rows, key=lambda o: -hist[o])}}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_opcode"}, indent=1))
for o in sorted(rows, key=lambda o: -hist[o]):
    print(f"{o:26s} {100 * hist[o] / total:5.1f} %   lone {rows[o][0]:6.2f}   simd {rows[o][1]:6.2f}")
