// lentil_lens_jit.h -- a solve kernel specialised for ANY lens table, at run time (round 5).
//
// The reference compiles all of its 44 lenses into the plugin: a generated `case <lens>: {...}` body per lens is spliced into
// switch(lensModel) (include/auto_generated_lens_includes/load_lt_sample_aperture.h:4-47, used at src/lentil.h:1308), so every
// lens runs as straight-line code.  This library ships two such specialisations (csrc/generated/, tools/gen_lens_code.py) and
// an interpreter (LdsLens, ~15 x slower) for every other table.  Here the generator itself is part of the library:
//   lens_jit_emit()   the C++ twin of tools/gen_lens_code.py's PrefetchEmitter / Emitter -- from the packed table
//                     lentil_hip_set_lens builds (base polynomials and their c * e derivatives, the same doubles the
//                     interpreter multiplies) to the source of a `Lens_rt` with eval_bw / transmittance: coefficients in a
//                     constant array in order of use, fetched eight at a time with hand-placed s_load_dwordx16 a block ahead,
//                     integer powers shared between terms, term = c * f(x) * f(y) * f(dx) * f(dy) * lambda^e left to right,
//                     terms summed in table order: the interpreter's and the oracle's operations in their order;
//   hiprtc            (bound at run time: dlopen, like RCCL) compiles solve_po_kernel<GenLens<Lens_rt>, ...> -- the four
//                     instances a pass can launch -- from the library's own kernel sources, which ride inside the .so
//                     (generated/embedded_sources.inc, written by __graft_entry__.build());
//   a cache           of code objects on disk, in a directory private to the caller, keyed by the table's hash and the hash of
//                     the sources, the flags, the compiler's version and the device's architecture; every file checksummed;
//   a thread          per compilation (owned by the table's entry, joined when the library goes away): lentil_hip_set_lens returns at once, passes run the interpreter until the code object is
//                     there (~15 s the first time a table is seen, milliseconds from the cache) and the compiled kernel after.
// Bit-identical results either way (tests/test_gpu_lens_jit.py); LENTIL_LENS_JIT=0 switches it off.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "lentil_device.h"

namespace lentil_jit {

// ---------------------------------------------------------------------------------------------------------------
// the emitter
// ---------------------------------------------------------------------------------------------------------------
static inline std::string hexfloat(double v) {
  char b[64];
  snprintf(b, sizeof b, "%a", v);
  return b;
}

struct PowerTable {
  std::set<std::pair<int, int>> have;
  std::vector<std::string> lines;
  static const char *var(int v) { static const char *n[4] = {"x", "y", "dx", "dy"}; return n[v]; }
  // the name of lens_ipow(var, e), e >= 2, with the recursion's intermediate powers emitted once (src/lens.h:226-233)
  std::string power(int v, int e) {
    const std::string name = std::string(var(v)) + "_" + std::to_string(e);
    if (have.count({v, e})) return name;
    if (e == 2) {
      lines.push_back("  const double " + name + " = " + var(v) + " * " + var(v) + ";");
    } else {
      const int h = e / 2;
      const std::string p2 = h == 1 ? std::string(var(v)) : power(v, h);
      if (e & 1) lines.push_back("  const double " + name + " = " + var(v) + " * " + p2 + " * " + p2 + ";");      // (x * p2) * p2
      else lines.push_back("  const double " + name + " = " + p2 + " * " + p2 + ";");
    }
    have.insert({v, e});
    return name;
  }
};

static inline void term_exponents(const lentil::DevTerm &t, int e[5]) {
  for (int v = 0; v < 4; ++v) e[v] = (int)((t.e >> (4 * v)) & 15u);
  e[4] = (int)((t.e >> 16) & 15u);
}

// eval_bw: the polynomials in the order the Newton iteration uses them, coefficient blocks of 8 prefetched one ahead
struct PrefetchEmitter {
  static constexpr int kChunk = 8;
  std::vector<double> &coef;
  size_t base;
  struct Term { int poly; std::vector<std::string> f; };
  std::vector<Term> terms;
  std::vector<std::pair<std::string, int>> targets;
  PowerTable pw;
  std::set<int> lp_used;
  explicit PrefetchEmitter(std::vector<double> &c) : coef(c), base(c.size()) {}
  void poly(const std::string &target, const lentil::DevTerm *t, int n) {
    const int pi = (int)targets.size();
    targets.push_back({target, n});
    for (int i = 0; i < n; ++i) {
      int e[5];
      term_exponents(t[i], e);
      for (int v = 0; v < 4; ++v) if (e[v] >= 2) (void)pw.power(v, e[v]);
      coef.push_back(t[i].c);
      Term tm;
      tm.poly = pi;
      for (int v = 0; v < 4; ++v) {
        if (e[v] == 1) tm.f.push_back(PowerTable::var(v));
        else if (e[v] >= 2) tm.f.push_back(pw.power(v, e[v]));
      }
      if (e[4] >= 1) { lp_used.insert(e[4]); tm.f.push_back("lp" + std::to_string(e[4])); }
      terms.push_back(tm);
    }
  }
  std::vector<std::string> lines() {
    while ((coef.size() - base) % kChunk) coef.push_back(0.0);       // whole blocks, plus one the last prefetch may touch
    for (int i = 0; i < kChunk; ++i) coef.push_back(0.0);
    std::vector<std::string> L;
    for (int e : lp_used) L.push_back("  const double lp" + std::to_string(e) + " = lp[" + std::to_string(e) + "];");
    L.insert(L.end(), pw.lines.begin(), pw.lines.end());
    L.push_back("  lentil_v8d ca, cb;");
    L.push_back("  LENTIL_SLOAD8(ca, C, " + std::to_string(base * 8) + ");");
    std::set<int> started;
    const int n_chunks = ((int)terms.size() + kChunk - 1) / kChunk;
    std::vector<int> touched;
    for (int j = 0; j < n_chunks; ++j) {
      const char *cur = (j % 2 == 0) ? "ca" : "cb", *nxt = (j % 2 == 0) ? "cb" : "ca";
      if (!touched.empty()) {
        std::string s = "  LENTIL_SWAIT_AFTER" + std::to_string(touched.size()) + "(" + cur;
        for (int t : touched) s += ", acc" + std::to_string(t);
        L.push_back(s + ");");
      } else {
        L.push_back(std::string("  LENTIL_SWAIT(") + cur + ");");
      }
      std::set<int> tset;
      const int lo = j * kChunk, hi = std::min((int)terms.size(), lo + kChunk);
      for (int i = lo; i < hi; ++i) tset.insert(terms[i].poly);
      touched.assign(tset.begin(), tset.end());
      if (j + 1 < n_chunks) L.push_back(std::string("  LENTIL_SLOAD8(") + nxt + ", C, " + std::to_string((base + (size_t)(j + 1) * kChunk) * 8) + ");");
      for (int i = lo; i < hi; ++i) {
        std::string expr = std::string(cur) + "[" + std::to_string(i - lo) + "]";
        for (const std::string &f : terms[i].f) expr += " * " + f;
        const std::string a = "acc" + std::to_string(terms[i].poly);
        if (started.count(terms[i].poly)) L.push_back("  " + a + " = " + a + " + " + expr + ";");
        else { L.push_back("  double " + a + " = " + expr + ";"); started.insert(terms[i].poly); }
      }
    }
    for (size_t pi = 0; pi < targets.size(); ++pi)
      L.push_back("  " + targets[pi].first + " = " + (targets[pi].second ? "acc" + std::to_string(pi) : std::string("0.0")) + ";");
    return L;
  }
};
// (a chunk touches at most three polynomials only if every polynomial has at least three terms: LENTIL_SWAIT_AFTERn exists for
// n = 1 .. 8, see the preamble below)

// transmittance: one polynomial, the compiler's own scalar loads, the coefficient pointer laundered every 12 terms
struct PlainEmitter {
  static constexpr int kChunk = 12;
  std::vector<double> &coef;
  PowerTable pw;
  std::vector<std::string> body;
  explicit PlainEmitter(std::vector<double> &c) : coef(c) {}
  void poly(const std::string &target, const lentil::DevTerm *t, int n) {
    if (!n) { body.push_back("  " + target + " = 0.0;"); return; }
    std::vector<std::string> parts;
    for (int i = 0; i < n; ++i) {
      int e[5];
      term_exponents(t[i], e);
      for (int v = 0; v < 4; ++v) if (e[v] >= 2) (void)pw.power(v, e[v]);
    }
    for (int i = 0; i < n; ++i) {
      int e[5];
      term_exponents(t[i], e);
      coef.push_back(t[i].c);
      std::string f = "C[" + std::to_string(coef.size() - 1) + "]";
      for (int v = 0; v < 4; ++v) {
        if (e[v] == 1) f += std::string(" * ") + PowerTable::var(v);
        else if (e[v] >= 2) f += " * " + pw.power(v, e[v]);
      }
      if (e[4] >= 1) f += " * lp[" + std::to_string(e[4]) + "]";
      parts.push_back(f);
    }
    std::string dep = "x";
    for (size_t i = 0; i < parts.size(); i += kChunk) {
      body.push_back("  asm volatile(\"\" : \"+s\"(C) : \"v\"(" + dep + "));");
      std::string sum;
      for (size_t k = i; k < std::min(parts.size(), i + kChunk); ++k) sum += (k == i ? "" : "\n      + ") + parts[k];
      if (i == 0) body.push_back("  double acc0 = " + sum + ";");
      else body.push_back("  acc0 = acc0\n      + " + sum + ";");
      dep = "acc0";
    }
    body.push_back("  " + target + " = acc0;");
  }
  std::vector<std::string> lines() {
    std::vector<std::string> L = pw.lines;
    L.insert(L.end(), body.begin(), body.end());
    return L;
  }
};

// the source of "lens_rt.h": namespace lentil::gen, struct Lens_rt (what csrc/generated/lens_<name>.h holds for a shipped lens)
static inline std::string lens_jit_emit(const lentil::DevLens &h, const std::vector<lentil::DevTerm> &terms, unsigned long long table_hash) {
  using namespace lentil;
  std::vector<double> coef;
  PrefetchEmitter em(coef);
  auto P = [&](int id) { return terms.data() + h.first[id]; };
  auto N = [&](int id) { return (int)h.count[id]; };
  em.poly("pred_ap[0]", P(P_AP_X), N(P_AP_X));
  em.poly("pred_ap[1]", P(P_AP_Y), N(P_AP_Y));
  { const int ids[4] = {P_DAP_00, P_DAP_01, P_DAP_10, P_DAP_11}; for (int i = 0; i < 4; ++i) em.poly("Jap[" + std::to_string(i) + "]", P(ids[i]), N(ids[i])); }
  { const int ids[4] = {P_OUT_X, P_OUT_Y, P_OUT_DX, P_OUT_DY}; for (int i = 0; i < 4; ++i) em.poly("out[" + std::to_string(i) + "]", P(ids[i]), N(ids[i])); }
  { const int ids[4] = {P_DOUT_00, P_DOUT_01, P_DOUT_10, P_DOUT_11}; for (int i = 0; i < 4; ++i) em.poly("Jout[" + std::to_string(i) + "]", P(ids[i]), N(ids[i])); }
  const std::vector<std::string> bw = em.lines();
  PlainEmitter et(coef);
  et.poly("const double t", P(P_OUT_T), N(P_OUT_T));
  const std::vector<std::string> tl = et.lines();
  std::string s;
  char hb[32];
  snprintf(hb, sizeof hb, "0x%016llx", table_hash);
  s += "// emitted by lentil_lens_jit.h (lens_jit_emit) for the table with hash " + std::string(hb) + "\n#pragma once\n";
  s += "#define LENTIL_COEF_PTR(NAME, ARR) \\\n  const __attribute__((address_space(4))) double *NAME = (const __attribute__((address_space(4))) double *)(ARR); \\\n  asm volatile(\"\" : \"+s\"(NAME))\n";
  s += "typedef double lentil_v8d __attribute__((ext_vector_type(8)));\n";
  s += "#define LENTIL_SLOAD8(DST, PTR, BYTES) asm volatile(\"s_load_dwordx16 %0, %1, \" #BYTES : \"=s\"(DST) : \"s\"(PTR))\n";
  s += "#define LENTIL_SWAIT(BLK) asm volatile(\"s_waitcnt lgkmcnt(0)\" : \"+s\"(BLK))\n";
  // (a block of eight terms may touch up to eight polynomials where polynomials are short)
  for (int n = 1; n <= 8; ++n) {
    std::string args, ops;
    for (int i = 0; i < n; ++i) { args += ", A" + std::to_string(i); ops += std::string(i ? ", " : "") + "\"v\"(A" + std::to_string(i) + ")"; }
    s += "#define LENTIL_SWAIT_AFTER" + std::to_string(n) + "(BLK" + args + ") asm volatile(\"s_waitcnt lgkmcnt(0)\" : \"+s\"(BLK) : " + ops + ")\n";
  }
  s += "namespace lentil { namespace gen {\n";
  s += "__device__ __constant__ double kCoef_rt[" + std::to_string(coef.size()) + "] = {\n";
  for (size_t i = 0; i < coef.size(); ++i) s += (i % 4 == 0 ? "    " : " ") + hexfloat(coef[i]) + "," + (i % 4 == 3 ? "\n" : "");
  s += "\n};\nstruct Lens_rt {\n";
  s += "  static constexpr unsigned long long kTableHash = " + std::string(hb) + "ull;\n";
  s += "  static __device__ __forceinline__ void eval_bw(const double v[4], const double *lp, double pred_ap[2],\n"
       "                                                 double Jap[4], double out[4], double Jout[4]) {\n"
       "  const double x = v[0], y = v[1], dx = v[2], dy = v[3];\n  LENTIL_COEF_PTR(C, kCoef_rt);\n";
  for (const std::string &l : bw) s += l + "\n";
  s += "  }\n  static __device__ __forceinline__ double transmittance(const double v[4], const double *lp) {\n"
       "  const double x = v[0], y = v[1], dx = v[2], dy = v[3];\n  LENTIL_COEF_PTR(C, kCoef_rt);\n";
  for (const std::string &l : tl) s += l + "\n";
  s += "  return t;\n  }\n};\n}}  // namespace lentil::gen\n";
  return s;
}

// ---------------------------------------------------------------------------------------------------------------
// hiprtc, bound at run time
// ---------------------------------------------------------------------------------------------------------------
struct Rtc {
  void *lib = nullptr;
  int (*CreateProgram)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*AddNameExpression)(void *, const char *) = nullptr;
  int (*CompileProgram)(void *, int, const char **) = nullptr;
  int (*GetProgramLogSize)(void *, size_t *) = nullptr;
  int (*GetProgramLog)(void *, char *) = nullptr;
  int (*GetLoweredName)(void *, const char *, const char **) = nullptr;
  int (*GetCodeSize)(void *, size_t *) = nullptr;
  int (*GetCode)(void *, char *) = nullptr;
  int (*DestroyProgram)(void **) = nullptr;
  int (*Version)(int *, int *) = nullptr;
  bool ok = false;
};
static inline Rtc &rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {getenv("LENTIL_HIPRTC_LIB"), "libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
    for (const char *n : names) {
      if (!n || !n[0]) continue;
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
#define LENTIL_RTC_SYM(F) *(void **)(&r.F) = dlsym(r.lib, "hiprtc" #F)
    LENTIL_RTC_SYM(CreateProgram); LENTIL_RTC_SYM(AddNameExpression); LENTIL_RTC_SYM(CompileProgram); LENTIL_RTC_SYM(GetProgramLogSize);
    LENTIL_RTC_SYM(GetProgramLog); LENTIL_RTC_SYM(GetLoweredName); LENTIL_RTC_SYM(GetCodeSize); LENTIL_RTC_SYM(GetCode); LENTIL_RTC_SYM(DestroyProgram); LENTIL_RTC_SYM(Version);
#undef LENTIL_RTC_SYM
    r.ok = r.CreateProgram && r.AddNameExpression && r.CompileProgram && r.GetProgramLogSize && r.GetProgramLog && r.GetLoweredName &&
           r.GetCodeSize && r.GetCode && r.DestroyProgram;
  });
  return r;
}

static inline uint64_t fnv(const void *d, size_t n, uint64_t h = 0xcbf29ce484222325ull) {
  const unsigned char *b = static_cast<const unsigned char *>(d);
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
  return h;
}

// the four instances of solve_po_kernel a pass launches for a compiled lens: [chroma][stream]
static const char *const kInstance[2][2] = {
    {"solve_po_kernel<lentil::GenLens<lentil::gen::Lens_rt>, false, false, false>", "solve_po_kernel<lentil::GenLens<lentil::gen::Lens_rt>, false, false, true>"},
    {"solve_po_kernel<lentil::GenLens<lentil::gen::Lens_rt>, false, true, false>", "solve_po_kernel<lentil::GenLens<lentil::gen::Lens_rt>, false, true, true>"}};

struct CodeObject {
  std::vector<char> code;
  std::string name[2][2];          // lowered (mangled) kernel names
};

struct Source { const char *name; const char *text; };

// sources: the library's own headers (embedded_sources.inc); lens_src: lens_jit_emit's output
// checks: lines appended to the translation unit (static_asserts that the structs the kernels share with the library have the
// library's own sizes: a kernel built from other sources or macros would otherwise agree with it by layout only)
static inline bool compile(const std::vector<Source> &sources, const std::string &lens_src, const std::vector<std::string> &flags,
                           CodeObject &out, std::string &log, const std::string &checks = std::string()) {
  Rtc &r = rtc();
  if (!r.ok) { log = "hiprtc is not available (libhiprtc.so)"; return false; }
  std::vector<std::string> texts;
  std::vector<const char *> names, bodies;
  static const char kStdint[] =
      "#pragma once\ntypedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
      "typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;\n";
  static const char kEmpty[] = "#pragma once\n";
  static const char kRegistry[] = "#pragma once\n#include \"lens_rt.h\"\n#define LENTIL_GENERATED_LENSES(X) X(rt)\n";
  texts.reserve(sources.size());
  for (const Source &s : sources) {
    std::string t = s.text;
    const std::string rel = "#include \"../../include/lentil_hip.h\"";       // (hiprtc's header names are flat)
    const size_t at = t.find(rel);
    if (at != std::string::npos) t.replace(at, rel.size(), "#include \"lentil_hip.h\"");
    texts.push_back(t);
    names.push_back(s.name);
  }
  for (const std::string &t : texts) bodies.push_back(t.c_str());
  names.push_back("lens_rt.h"); bodies.push_back(lens_src.c_str());
  names.push_back("generated/lens_registry.h"); bodies.push_back(kRegistry);
  names.push_back("stdint.h"); bodies.push_back(kStdint);
  names.push_back("stddef.h"); bodies.push_back(kEmpty);
  names.push_back("hip/hip_runtime.h"); bodies.push_back(kEmpty);
  std::string tu = "#include \"lentil_kernels.h\"\n" + checks;
  for (int c = 0; c < 2; ++c)
    for (int s = 0; s < 2; ++s) tu += std::string("template __global__ void ") + kInstance[c][s] + "(DrawArgs);\n";
  void *prog = nullptr;
  if (r.CreateProgram(&prog, tu.c_str(), "lentil_lens_rt.hip", (int)names.size(), bodies.data(), names.data()) != 0) { log = "hiprtcCreateProgram failed"; return false; }
  for (int c = 0; c < 2; ++c)
    for (int s = 0; s < 2; ++s) (void)r.AddNameExpression(prog, kInstance[c][s]);
  std::vector<const char *> opts;
  for (const std::string &f : flags) opts.push_back(f.c_str());
  const int rc = r.CompileProgram(prog, (int)opts.size(), opts.data());
  size_t n = 0;
  if (r.GetProgramLogSize(prog, &n) == 0 && n > 1) { log.resize(n); (void)r.GetProgramLog(prog, &log[0]); }
  bool ok = rc == 0;
  if (ok) {
    for (int c = 0; c < 2 && ok; ++c)
      for (int s = 0; s < 2 && ok; ++s) {
        const char *low = nullptr;
        ok = r.GetLoweredName(prog, kInstance[c][s], &low) == 0 && low;
        if (ok) out.name[c][s] = low;
      }
    size_t cs = 0;
    ok = ok && r.GetCodeSize(prog, &cs) == 0 && cs > 0;
    if (ok) { out.code.resize(cs); ok = r.GetCode(prog, out.code.data()) == 0; }
    if (!ok && log.empty()) log = "hiprtc: no code object";
  }
  (void)r.DestroyProgram(&prog);
  return ok;
}

// ---- the cache on disk: <dir>/<table hash>_<source hash>.lco = "LCO2", u64 checksum, 4 x (u32 length, name), u64 size, code ------
// The directory is the caller's own and nobody else's: LENTIL_JIT_CACHE, $XDG_CACHE_HOME/lentil_hip, $HOME/.cache/lentil_hip, or
// /tmp/lentil_hip_<uid>; created 0700, and used only if it is a directory (not a link) that the caller owns and that neither its
// group nor others can write to.  Without such a directory there is no disk cache (every process compiles for itself).  A file is
// loaded only if it is a regular file with the same owner and permissions and its checksum (FNV-1a over names and code) holds:
// what is read here is handed to hipModuleLoadData and runs on the GPU inside the render.
static inline std::string cache_dir_wanted() {
  if (const char *e = getenv("LENTIL_JIT_CACHE")) if (e[0]) return e;
  if (const char *e = getenv("XDG_CACHE_HOME")) if (e[0]) return std::string(e) + "/lentil_hip";
  if (const char *e = getenv("HOME")) if (e[0]) return std::string(e) + "/.cache/lentil_hip";
  return "/tmp/lentil_hip_" + std::to_string((unsigned long long)geteuid());
}
static inline bool private_to_caller(const struct stat &st) { return st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0; }
// the directory if it can be trusted (created on the way when `create`), "" otherwise
static inline std::string cache_dir(bool create) {
  const std::string p = cache_dir_wanted();
  if (p.empty()) return "";
  if (create)
    for (size_t i = 1; i <= p.size(); ++i)
      if (i == p.size() || p[i] == '/') (void)mkdir(p.substr(0, i).c_str(), 0700);      // (parents that exist keep their modes)
  struct stat st;
  if (lstat(p.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || !private_to_caller(st)) return "";
  return p;
}
static inline std::string cache_file(const std::string &dir, uint64_t table_hash, uint64_t source_hash) {
  if (dir.empty()) return "";
  char b[64];
  snprintf(b, sizeof b, "/%016llx_%016llx.lco", (unsigned long long)table_hash, (unsigned long long)source_hash);
  return dir + b;
}
static inline uint64_t code_checksum(const CodeObject &co) {
  uint64_t h = fnv("lco2", 4);
  for (int c = 0; c < 2; ++c)
    for (int s = 0; s < 2; ++s) h = fnv(co.name[c][s].data(), co.name[c][s].size(), h);
  return fnv(co.code.data(), co.code.size(), h);
}
static inline bool cache_load(const std::string &path, CodeObject &out) {
  if (path.empty()) return false;
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || !private_to_caller(st)) { close(fd); return false; }
  FILE *f = fdopen(fd, "rb");
  if (!f) { close(fd); return false; }
  bool ok = false;
  char magic[4];
  uint64_t sum = 0;
  if (fread(magic, 1, 4, f) == 4 && memcmp(magic, "LCO2", 4) == 0 && fread(&sum, 8, 1, f) == 1) {
    ok = true;
    for (int c = 0; c < 2 && ok; ++c)
      for (int s = 0; s < 2 && ok; ++s) {
        uint32_t n = 0;
        ok = fread(&n, 4, 1, f) == 1 && n > 0 && n < 4096;
        if (ok) { out.name[c][s].resize(n); ok = fread(&out.name[c][s][0], 1, n, f) == n; }
      }
    uint64_t cs = 0;
    ok = ok && fread(&cs, 8, 1, f) == 1 && cs > 0 && cs < (1ull << 30);
    if (ok) { out.code.resize(cs); ok = fread(out.code.data(), 1, cs, f) == cs; }
    ok = ok && code_checksum(out) == sum;
  }
  fclose(f);
  if (!ok) { out.code.clear(); for (int c = 0; c < 2; ++c) for (int s = 0; s < 2; ++s) out.name[c][s].clear(); }
  return ok;
}
static inline void cache_store(const std::string &path, const CodeObject &co) {
  if (path.empty()) return;
  const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  if (fd < 0) return;
  FILE *f = fdopen(fd, "wb");
  if (!f) { close(fd); (void)remove(tmp.c_str()); return; }
  const uint64_t sum = code_checksum(co);
  bool ok = fwrite("LCO2", 1, 4, f) == 4 && fwrite(&sum, 8, 1, f) == 1;
  for (int c = 0; c < 2 && ok; ++c)
    for (int s = 0; s < 2 && ok; ++s) {
      const uint32_t n = (uint32_t)co.name[c][s].size();
      ok = fwrite(&n, 4, 1, f) == 1 && fwrite(co.name[c][s].data(), 1, n, f) == n;
    }
  const uint64_t cs = co.code.size();
  ok = ok && fwrite(&cs, 8, 1, f) == 1 && fwrite(co.code.data(), 1, cs, f) == cs;
  ok = (fclose(f) == 0) && ok;
  if (ok) (void)rename(tmp.c_str(), path.c_str());       // (atomic: a reader sees the whole file or none)
  else (void)remove(tmp.c_str());
}

// ---- one specialisation per table hash, shared by the contexts of a process -------------------------------------------------
struct Entry {
  enum State { kCompiling = 1, kReady = 2, kFailed = -1 };
  std::atomic<int> state{kCompiling};
  CodeObject co;
  std::string log;
  double seconds = 0.0;
  bool from_cache = false;
  std::thread worker;       // the compilation, if one was started: joined by join_all() (library teardown) -- never detached
};
// (both live for as long as the library is mapped and are never destroyed: the teardown hook below walks them after the
// process's static destructors may already have run)
static inline std::mutex &registry_mutex() { static std::mutex *m = new std::mutex; return *m; }
static inline std::map<uint64_t, std::shared_ptr<Entry>> &registry() { static auto *r = new std::map<uint64_t, std::shared_ptr<Entry>>; return *r; }
// Waits for every compilation still running.  Called when the library goes away (a destructor function of the .so: process
// exit and dlclose) -- a compiling thread is inside libhiprtc / comgr and this library's own code, neither of which may be
// unmapped or torn down under it.  hiprtc has no way to cancel a compilation; the wait is its remaining seconds.
static inline void join_all() {
  std::vector<std::shared_ptr<Entry>> all;
  {
    std::lock_guard<std::mutex> lock(registry_mutex());
    for (auto &kv : registry()) all.push_back(kv.second);
  }
  for (auto &e : all)
    if (e->worker.joinable() && e->worker.get_id() != std::this_thread::get_id()) e->worker.join();
}

}  // namespace lentil_jit
