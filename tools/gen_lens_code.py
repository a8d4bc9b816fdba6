#!/usr/bin/env python3
"""Generate straight-line HIP code for the shipped lens tables.

The reference compiles each lens's polynomials into the plugin (generated `case <lens>: {...}`
bodies spliced into switch(lensModel), src/lentil.h:1262,1278,1308) -- the gfx950 analogue is a
kernel instantiation per shipped lens: coefficients become literals (SGPR pairs), integer powers
are shared between terms, nothing is read from memory.  The arithmetic is *exactly* the table
interpreter's / oracle's: term = c * f(x) * f(y) * f(dx) * f(dy) * f(lambda) left to right with
f(v) = v or lens_ipow(v, e) (src/lens.h:226-233 recursion), terms summed in table order; the build
uses -ffp-contract=off, so the compiler may share sub-expressions but not re-round anything.

Usage: python tools/gen_lens_code.py      (reads pota_amd/lenses/*.json, writes
       pota_amd/csrc/generated/lens_<name>.h and lens_registry.h)
"""
import hashlib
import json
import os
import struct

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LENS_DIR = os.path.join(ROOT, "pota_amd", "lenses")
OUT_DIR = os.path.join(ROOT, "pota_amd", "csrc", "generated")
VARS = ["x", "y", "dx", "dy"]
OUT_NAMES = ["out_x", "out_y", "out_dx", "out_dy", "out_t"]
AP_NAMES = ["ap_x", "ap_y", "ap_dx", "ap_dy"]


def derive(terms, var):
    return [(float(c) * float(e[var]), [ee - (1 if i == var else 0) for i, ee in enumerate(e)])
            for c, e in terms if e[var] > 0]


def table_hash(spec):
    """FNV-1a over (coefficient bits, exponents) of the 9 base polynomials in ABI order -- the same
    hash liblentil_hip computes in lentil_hip_set_lens to pick a compiled-in specialisation."""
    h = 0xCBF29CE484222325
    for name in OUT_NAMES + AP_NAMES:
        for c, e in spec["polys"][name]:
            data = struct.pack("<d5B", float(c), *[int(v) for v in e])
            for b in data:
                h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        h = ((h ^ 0xFF) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


class Emitter:
    def __init__(self, coef, cname):
        self.lines = []
        self.pows = {}     # (var, e) -> name
        self.coef = coef   # shared coefficient list (constant memory, read with scalar loads)
        self.cname = cname
        self.nacc = 0
        self.dep = "x"     # value the next coefficient-pointer fence depends on
    CHUNK = 12

    def power(self, var, e):
        """Name of lens_ipow(var, e), e >= 2, emitting the recursion's intermediate powers once."""
        key = (var, e)
        if key in self.pows:
            return self.pows[key]
        v = VARS[var]
        name = "%s_%d" % (v, e)
        if e == 2:
            self.lines.append("  const double %s = %s * %s;" % (name, v, v))
        else:
            h = e // 2
            p2 = v if h == 1 else self.power(var, h)
            if e & 1:
                self.lines.append("  const double %s = %s * %s * %s;" % (name, v, p2, p2))   # (x*p2)*p2
            else:
                self.lines.append("  const double %s = %s * %s;" % (name, p2, p2))
        self.pows[key] = name
        return name

    def poly(self, target, terms):
        if not terms:
            self.lines.append("  %s = 0.0;" % target)
            return
        # make sure powers exist before the expression
        for c, e in terms:
            for var in range(4):
                if e[var] >= 2:
                    self.power(var, e[var])
        parts = []
        for c, e in terms:
            self.coef.append(float(c))
            f = ["%s[%d]" % (self.cname, len(self.coef) - 1)]
            for var in range(4):
                if e[var] == 1:
                    f.append(VARS[var])
                elif e[var] >= 2:
                    f.append(self.pows[(var, e[var])])
            if e[4] >= 1:
                f.append("lp[%d]" % e[4])
            parts.append(" * ".join(f))
        # Sum left to right in chunks of CHUNK terms.  Before each chunk the coefficient pointer is
        # laundered through an empty asm that also *consumes* the value computed just before, so the
        # chunk's scalar loads can neither be hoisted out of the solver loop nor be clustered at the top of
        # the iteration (either would need ~800 live SGPRs and spill them through VGPR lanes).
        tmp = "acc%d" % self.nacc
        self.nacc += 1
        for i in range(0, len(parts), self.CHUNK):
            chunk = parts[i:i + self.CHUNK]
            self.lines.append("  asm volatile(\"\" : \"+s\"(C) : \"v\"(%s));" % self.dep)
            if i == 0:
                self.lines.append("  double %s = %s;" % (tmp, "\n      + ".join(chunk)))
            else:
                self.lines.append("  %s = %s\n      + %s;" % (tmp, tmp, "\n      + ".join(chunk)))
            self.dep = tmp
        self.lines.append("  %s = %s;" % (target, tmp))


class PrefetchEmitter:
    """eval_bw with its coefficient loads written out: `s_load_dwordx16` of the next 8 coefficients into the other of
    two SGPR blocks is issued BEFORE the current 8 terms are computed, and waited for (lgkmcnt(0): scalar loads return out
    of order, so that is the only count that means anything) only when its terms come up.  The compiler's own placement
    (load, wait at once, compute) exposes the scalar cache's latency 60 times per Newton iteration -- what one wave
    alone on a SIMD spends half its time on.  The lambda powers are read from LDS once, ahead of the first load, so that
    no LDS wait of the compiler's falls between a prefetch and its use."""
    CHUNK = 8

    def __init__(self, coef):
        self.coef = coef
        self.base = len(coef)
        self.terms = []          # (poly index, expression with C? placeholder)
        self.targets = []
        self.pows = {}
        self.pow_lines = []
        self.lp_used = set()

    def power(self, var, e):
        key = (var, e)
        if key in self.pows:
            return self.pows[key]
        v = VARS[var]
        name = "%s_%d" % (v, e)
        if e == 2:
            self.pow_lines.append("  const double %s = %s * %s;" % (name, v, v))
        else:
            h = e // 2
            p2 = v if h == 1 else self.power(var, h)
            if e & 1:
                self.pow_lines.append("  const double %s = %s * %s * %s;" % (name, v, p2, p2))
            else:
                self.pow_lines.append("  const double %s = %s * %s;" % (name, p2, p2))
        self.pows[key] = name
        return name

    def poly(self, target, terms):
        pi = len(self.targets)
        self.targets.append((target, len(terms)))
        for c, e in terms:
            for var in range(4):
                if e[var] >= 2:
                    self.power(var, e[var])
            self.coef.append(float(c))
            f = []
            for var in range(4):
                if e[var] == 1:
                    f.append(VARS[var])
                elif e[var] >= 2:
                    f.append(self.pows[(var, e[var])])
            if e[4] >= 1:
                self.lp_used.add(e[4])
                f.append("lp%d" % e[4])
            self.terms.append((pi, f))

    def lines(self):
        # pad the coefficient array: whole blocks, plus one the last prefetch may touch
        while (len(self.coef) - self.base) % self.CHUNK:
            self.coef.append(0.0)
        for _ in range(self.CHUNK):
            self.coef.append(0.0)
        L = []
        for e in sorted(self.lp_used):
            L.append("  const double lp%d = lp[%d];" % (e, e))
        L.extend(self.pow_lines)
        L.append("  lentil_v8d ca, cb;")
        L.append("  LENTIL_SLOAD8(ca, C, %d);" % (self.base * 8))
        started = set()
        n_chunks = (len(self.terms) + self.CHUNK - 1) // self.CHUNK
        touched = []
        for j in range(n_chunks):
            cur, nxt = ("ca", "cb") if j % 2 == 0 else ("cb", "ca")
            # (the wait also consumes what the chunk before it computed: the compiler may not push those terms behind
            # this point -- it would keep block after block of coefficients alive and spill them through VGPR lanes)
            if touched:
                L.append("  LENTIL_SWAIT_AFTER%d(%s, %s);" % (len(touched), cur, ", ".join("acc%d" % t for t in touched)))
            else:
                L.append("  LENTIL_SWAIT(%s);" % cur)
            touched = sorted(set(pi for pi, f in self.terms[j * self.CHUNK:(j + 1) * self.CHUNK]))
            if j + 1 < n_chunks:
                L.append("  LENTIL_SLOAD8(%s, C, %d);" % (nxt, (self.base + (j + 1) * self.CHUNK) * 8))
            for i, (pi, f) in enumerate(self.terms[j * self.CHUNK:(j + 1) * self.CHUNK]):
                expr = " * ".join(["%s[%d]" % (cur, i)] + f)
                if pi in started:
                    L.append("  acc%d = acc%d + %s;" % (pi, pi, expr))
                else:
                    L.append("  double acc%d = %s;" % (pi, expr))
                    started.add(pi)
        for pi, (target, n) in enumerate(self.targets):
            L.append("  %s = %s;" % (target, ("acc%d" % pi) if n else "0.0"))
        return L


def gen_lens(name, spec):
    polys = spec["polys"]
    dap = [[derive(polys["ap_" + a], 2 + j) for j in range(2)] for a in ("x", "y")]
    dout = [[derive(polys["out_" + a], j) for j in range(2)] for a in ("dx", "dy")]
    coef = []
    cname = "kCoef_%s" % name
    prefetch = os.environ.get("LENTIL_GEN_PREFETCH", "1") != "0"
    em = PrefetchEmitter(coef) if prefetch else Emitter(coef, "C")
    em.poly("pred_ap[0]", polys["ap_x"])
    em.poly("pred_ap[1]", polys["ap_y"])
    for i in range(2):
        for j in range(2):
            em.poly("Jap[%d]" % (i * 2 + j), dap[i][j])
    for i, n in enumerate(["out_x", "out_y", "out_dx", "out_dy"]):
        em.poly("out[%d]" % i, polys[n])
    for i in range(2):
        for j in range(2):
            em.poly("Jout[%d]" % (i * 2 + j), dout[i][j])
    em_lines = em.lines() if prefetch else em.lines
    et = Emitter(coef, "C")
    et.poly("const double t", polys["out_t"])
    n_terms = sum(len(polys[n]) for n in OUT_NAMES + AP_NAMES)
    h = table_hash(spec)
    src = []
    src.append("// GENERATED by tools/gen_lens_code.py from pota_amd/lenses/%s.json -- do not edit." % name)
    src.append("// Straight-line evaluation of the lens polynomials; same operation order as the table")
    src.append("// interpreter (LdsLens::eval) and the oracle.  %d base terms, table hash 0x%016x." % (n_terms, h))
    src.append("#pragma once")
    src.append("#ifndef LENTIL_COEF_PTR")
    src.append("// A constant-address-space pointer laundered through an empty asm: the coefficient loads stay")
    src.append("// scalar (s_load) but are not hoisted out of the solver loop (which would spill ~800 SGPRs).")
    src.append("#define LENTIL_COEF_PTR(NAME, ARR) \\")
    src.append("  const __attribute__((address_space(4))) double *NAME = (const __attribute__((address_space(4))) double *)(ARR); \\")
    src.append("  asm volatile(\"\" : \"+s\"(NAME))")
    src.append("#endif")
    src.append("#ifndef LENTIL_SLOAD8")
    src.append("// eight coefficients = 16 SGPRs, loaded by hand (PrefetchEmitter in tools/gen_lens_code.py says why); the wait takes the")
    src.append("// block as an in/out operand, so that no use of it can be scheduled ahead of the wait")
    src.append("typedef double lentil_v8d __attribute__((ext_vector_type(8)));")
    src.append("#define LENTIL_SLOAD8(DST, PTR, BYTES) asm volatile(\"s_load_dwordx16 %0, %1, \" #BYTES : \"=s\"(DST) : \"s\"(PTR))")
    src.append("#define LENTIL_SWAIT(BLK) asm volatile(\"s_waitcnt lgkmcnt(0)\" : \"+s\"(BLK))")
    src.append("#define LENTIL_SWAIT_AFTER1(BLK, A) asm volatile(\"s_waitcnt lgkmcnt(0)\" : \"+s\"(BLK) : \"v\"(A))")
    src.append("#define LENTIL_SWAIT_AFTER2(BLK, A, B) asm volatile(\"s_waitcnt lgkmcnt(0)\" : \"+s\"(BLK) : \"v\"(A), \"v\"(B))")
    src.append("#define LENTIL_SWAIT_AFTER3(BLK, A, B, D) asm volatile(\"s_waitcnt lgkmcnt(0)\" : \"+s\"(BLK) : \"v\"(A), \"v\"(B), \"v\"(D))")
    src.append("#endif")
    src.append("namespace lentil { namespace gen {")
    src.append("// coefficients in order of use (base terms and c*e derivative terms); constant address space,")
    src.append("// uniform indices -> the compiler fetches them with wide scalar loads (s_load_dwordx8/x16)")
    src.append("__device__ __constant__ double %s[%d] = {" % (cname, len(coef)))
    for i in range(0, len(coef), 4):
        src.append("    " + " ".join("%s," % v.hex() for v in coef[i:i + 4]))
    src.append("};")
    src.append("struct Lens_%s {" % name)
    src.append("  static constexpr unsigned long long kTableHash = 0x%016xull;" % h)
    src.append("  // v = (x, y, dx, dy); lp[e] = lens_ipow(lambda, e)")
    src.append("  static __device__ __forceinline__ void eval_bw(const double v[4], const double *lp, double pred_ap[2],")
    src.append("                                                 double Jap[4], double out[4], double Jout[4]) {")
    src.append("  const double x = v[0], y = v[1], dx = v[2], dy = v[3];")
    src.append("  LENTIL_COEF_PTR(C, %s);" % cname)
    src.extend(em_lines)
    src.append("  }")
    src.append("  static __device__ __forceinline__ double transmittance(const double v[4], const double *lp) {")
    src.append("  const double x = v[0], y = v[1], dx = v[2], dy = v[3];")
    src.append("  LENTIL_COEF_PTR(C, %s);" % cname)
    src.extend(et.lines)
    src.append("  return t;")
    src.append("  }")
    src.append("};")
    src.append("}}  // namespace lentil::gen")
    with open(os.path.join(OUT_DIR, "lens_%s.h" % name), "w") as f:
        f.write("\n".join(src) + "\n")
    return h


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    names = sorted(f[:-5] for f in os.listdir(LENS_DIR) if f.endswith(".json"))
    only = os.environ.get("LENTIL_GEN_LENSES", "double_gauss_50mm,petzval_58mm")     # (every compiled lens is four more solve kernels to build)
    names = [n for n in names if n in only.split(",")]
    reg = ["// GENERATED by tools/gen_lens_code.py -- the lenses compiled into liblentil_hip.so.",
           "#pragma once"]
    for n in names:
        with open(os.path.join(LENS_DIR, n + ".json")) as f:
            spec = json.load(f)
        h = gen_lens(n, spec)
        reg.append('#include "lens_%s.h"' % n)
        print("generated lens_%s.h (hash %016x)" % (n, h))
    reg.append("#define LENTIL_GENERATED_LENSES(X) " + " ".join("X(%s)" % n for n in names))
    with open(os.path.join(OUT_DIR, "lens_registry.h"), "w") as f:
        f.write("\n".join(reg) + "\n")


if __name__ == "__main__":
    main()
