// Run-time compilation of generated kernels (fused_codegen.h, lbfgs_codegen.h, wave_codegen.h): hiprtc (or, on request, the
// ROCm install's clang++ — "which compiler" below) -> code object for gfx950 -> hipModuleLoadData.  Code objects
// are cached on disk by the hash of their source and the compiler's identity ($DNLP_KERNEL_CACHE, else $XDG_CACHE_HOME/dnlp_kernel_cache, else
// /tmp/dnlp_kernel_cache-<uid>), so a problem structure is compiled once per machine.  The cache
// directory is used only when it is a real directory owned by this user with mode 0700 (code objects
// are executed: a directory somebody else could have planted is ignored and the kernel is compiled
// afresh); a cached object that no longer loads is deleted and recompiled.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <limits.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern char** environ;

namespace dnlp {

inline uint64_t rtc_hash(const std::string& s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
  return h;
}

inline std::string rtc_cache_dir() {
  if (const char* d = std::getenv("DNLP_KERNEL_CACHE")) return d;
  if (const char* x = std::getenv("XDG_CACHE_HOME")) if (*x) return std::string(x) + "/dnlp_kernel_cache";
  return "/tmp/dnlp_kernel_cache-" + std::to_string(static_cast<long>(getuid()));
}

// The cache directory, created if absent; "" when it cannot be trusted (not a directory, a symlink,
// another owner, or group / world accessible).
inline std::string rtc_trusted_cache_dir() {
  const std::string dir = rtc_cache_dir();
  struct stat st;
  if (lstat(dir.c_str(), &st) != 0) {
    if (mkdir(dir.c_str(), 0700) != 0 || lstat(dir.c_str(), &st) != 0) return "";
  }
  if (!S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & 077) != 0) return "";
  return dir;
}

// ---- which compiler ------------------------------------------------------------------------------------------------------
// In-process hiprtc is whatever libhiprtc / libamd_comgr the loader bound first: a process that imported PyTorch compiles with
// the pair torch ships in torch/lib (ROCm 7.0 in this image), any other with the ROCm install's (7.2).  Both answer
// hiprtcVersion alike; their code objects differ.  The 7.0 compiler gives every module that CALLS a function (the wavefront
// solver's kernels: wave_ipm.h keeps its phases out of line) half the register budget as accumulation registers — 128 + 128
// under a bound of 256, spills moved through v_accvgpr — where 7.2 proves that nothing needs them and allocates 256 ordinary
// ones (and 7.0 let a kernel entry point with a body of its own take registers on top of that split: wave_wg_kernel.h).
// Measured on the MI355X in the same bench.py command (torch imported), 7.0 against 7.2: path planning 14.00 / 14.08 k
// problems/s, power flow 14.14 / 14.48 k at 1024 instances, localization 431.1 / 434.5 k at 8192 — inside 2.5 %, so the default
// stays the in-process compiler whichever it is.  $DNLP_RTC_COMPILER=clang compiles through the install's clang++ instead
// (<ROCm root>/lib/llvm/bin/clang++ or $DNLP_RTC_CLANG, run as a child process): the same code objects whether or not
// torch was imported first.  The compiler's identity (path, size, time of its file) is part of the cache key: the two must
// not share entries.
inline std::string rtc_file_identity(const std::string& path) {
  struct stat st;
  if (path.empty() || stat(path.c_str(), &st) != 0) return path + "#?";
  return path + "#" + std::to_string(static_cast<long long>(st.st_size)) + "#" + std::to_string(static_cast<long long>(st.st_mtime));
}

inline std::string rtc_real(const std::string& path) {
  char buf[PATH_MAX];
  return realpath(path.c_str(), buf) ? std::string(buf) : path;
}

inline std::string rtc_bound_hiprtc() {
  Dl_info di;
  if (!dladdr(reinterpret_cast<const void*>(&hiprtcCompileProgram), &di) || !di.dli_fname) return "";
  return rtc_real(di.dli_fname);
}

inline std::string rtc_rocm_root() {
  const char* r = std::getenv("ROCM_PATH");
  return rtc_real(r && *r ? r : "/opt/rocm");
}

// the install's offline compiler ("" when there is none): $DNLP_RTC_CLANG, else <ROCm root>/lib/llvm/bin/clang++
inline std::string rtc_install_clang() {
  const char* e = std::getenv("DNLP_RTC_CLANG");
  const std::string c = e && *e ? std::string(e) : rtc_rocm_root() + "/lib/llvm/bin/clang++";
  return access(c.c_str(), X_OK) == 0 ? c : std::string();
}

struct RtcChoice {
  std::string clang;        // non-empty: compile with this clang++ as a child process
  std::string identity;     // of the compiler taken (cache key)
};

inline RtcChoice rtc_choice() {
  RtcChoice c;
  const char* e = std::getenv("DNLP_RTC_COMPILER");
  if (e && std::string(e) == "clang") c.clang = rtc_install_clang();
  c.identity = rtc_file_identity(c.clang.empty() ? rtc_bound_hiprtc() : c.clang);
  return c;
}

inline std::string rtc_cache_path(const std::string& src, const std::string& arch, const RtcChoice& c) {
  const std::string dir = rtc_trusted_cache_dir();
  if (dir.empty()) return "";
  int major = 0, minor = 0;
  hiprtcVersion(&major, &minor);
  char name[64];
  std::snprintf(name, sizeof name, "%016llx.hsaco",
                static_cast<unsigned long long>(rtc_hash(src + arch + "#k3#hiprtc" + std::to_string(major) + "." +
                                                         std::to_string(minor) + "#" + c.identity)));
  return dir + "/" + name;
}

// forget the cached object of `src` (it loaded, but the caller cannot launch it)
inline void rtc_cache_drop(const std::string& src) {
  const std::string path = rtc_cache_path(src, "--offload-arch=gfx950", rtc_choice());
  if (!path.empty()) std::remove(path.c_str());
}

inline bool rtc_read_file(const std::string& path, std::vector<char>& out) {
  out.clear();
  FILE* fp = std::fopen(path.c_str(), "rb");
  if (!fp) return false;
  std::fseek(fp, 0, SEEK_END);
  const long n = std::ftell(fp);
  std::fseek(fp, 0, SEEK_SET);
  out.resize(static_cast<size_t>(n > 0 ? n : 0));
  const size_t got = out.empty() ? 0 : std::fread(out.data(), 1, out.size(), fp);
  std::fclose(fp);
  if (got != out.size()) { out.clear(); return false; }
  return true;
}

inline std::atomic<unsigned>& rtc_writer() { static std::atomic<unsigned> w{0}; return w; }

// `src` through the install's clang++ (a child process: posix_spawn, the parent's environment without LD_PRELOAD — a profiler's
// preloaded library has no business in the compiler).  What hiprtc includes by itself is named here: hip/hip_runtime.h.
inline std::vector<char> rtc_compile_child(const std::string& clang, const std::string& src, std::string& log) {
  std::vector<char> code;
  std::string dir = rtc_trusted_cache_dir();
  std::string made;
  if (dir.empty()) {
    char tmpl[] = "/tmp/dnlp_rtc_XXXXXX";
    if (!mkdtemp(tmpl)) { log = "no directory to compile in"; return code; }
    dir = made = tmpl;
  }
  const std::string base = dir + "/build." + std::to_string(static_cast<long>(getpid())) + "." + std::to_string(rtc_writer().fetch_add(1));
  const std::string f_src = base + ".hip", f_out = base + ".hsaco", f_log = base + ".log";
  bool ok = false;
  if (FILE* fp = std::fopen(f_src.c_str(), "wb")) {
    ok = std::fwrite(src.data(), 1, src.size(), fp) == src.size();
    std::fclose(fp);
  }
  if (ok) {
    const std::string rocm = "--rocm-path=" + rtc_rocm_root();
    std::vector<std::string> av = {clang, "-x", "hip", "--offload-arch=gfx950", "--cuda-device-only", "--no-gpu-bundle-output", "-O3", "-std=c++17",
                                   "-munsafe-fp-atomics", rocm, "-include", "hip/hip_runtime.h", "-o", f_out, f_src};
    std::vector<char*> argv;
    for (auto& a : av) argv.push_back(&a[0]);
    argv.push_back(nullptr);
    std::vector<char*> envp;
    for (char** e = environ; e && *e; ++e) if (std::strncmp(*e, "LD_PRELOAD=", 11) != 0) envp.push_back(*e);
    envp.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    posix_spawn_file_actions_addopen(&fa, 1, f_log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, clang.c_str(), &fa, nullptr, argv.data(), envp.data());
    posix_spawn_file_actions_destroy(&fa);
    int status = -1;
    if (rc == 0) { while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {} }
    std::vector<char> text;
    if (rtc_read_file(f_log, text)) log.assign(text.begin(), text.end());
    if (rc != 0) log = "posix_spawn(" + clang + ") failed";
    else if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { if (log.empty()) log = clang + " failed"; }
    else rtc_read_file(f_out, code);
  } else log = "cannot write " + f_src;
  std::remove(f_src.c_str()); std::remove(f_out.c_str()); std::remove(f_log.c_str());
  if (!made.empty()) rmdir(made.c_str());
  return code;
}

// Compile `src` for gfx950.  Returns the code object (empty on failure, `log` has the compiler text).
// One compilation at a time per process (the slots of a batch stream compile the same source on their own threads at the same
// time: behind the lock the second finds the first one's object in the cache).
inline std::mutex& rtc_compile_lock() { static std::mutex m; return m; }

inline std::vector<char> rtc_compile(const std::string& src, std::string& log, bool use_cache = true) {
  std::lock_guard<std::mutex> one(rtc_compile_lock());
  const std::string arch = "--offload-arch=gfx950";
  const RtcChoice choice = rtc_choice();
  const std::string path = use_cache ? rtc_cache_path(src, arch, choice) : std::string();
  if (path.empty()) use_cache = false;
  std::vector<char> code;
  if (use_cache && rtc_read_file(path, code) && !code.empty()) return code;
  code.clear();
  if (!choice.clang.empty()) {
    code = rtc_compile_child(choice.clang, src, log);
    if (code.empty()) std::fprintf(stderr, "[dnlp] %s did not compile a generated kernel (in-process hiprtc is used): %s\n", choice.clang.c_str(), log.substr(0, 600).c_str());
  }
  if (code.empty()) {
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "dnlp_generated.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
      log = "hiprtcCreateProgram failed";
      return code;
    }
    const char* opts[] = {arch.c_str(), "-O3", "-std=c++17", "-munsafe-fp-atomics"};
    const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) { log.resize(ls); hiprtcGetProgramLog(prog, &log[0]); }
    if (r != HIPRTC_SUCCESS) {
      if (log.empty()) log = hiprtcGetErrorString(r);
      hiprtcDestroyProgram(&prog);
      return code;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
  }
  if (use_cache && !code.empty()) {
    // (a name of its own per writer, then rename: another process may be writing the same entry)
    const std::string tmp = path + ".tmp" + std::to_string(static_cast<long>(getpid())) + "." + std::to_string(rtc_writer().fetch_add(1));
    if (FILE* fp = std::fopen(tmp.c_str(), "wb")) {
      const size_t w = std::fwrite(code.data(), 1, code.size(), fp);
      std::fclose(fp);
      if (w == code.size()) std::rename(tmp.c_str(), path.c_str()); else std::remove(tmp.c_str());
    }
  }
  return code;
}

struct RtcKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  bool tried = false, ok = false;
  std::string log;
  double compile_seconds = 0.0;
  ~RtcKernel() { if (mod) hipModuleUnload(mod); }
  bool load(const std::string& src, const char* entry) {
    tried = true;
    std::vector<char> code = rtc_compile(src, log);
    if (code.empty()) return false;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
      // a truncated / stale cached object: drop it and compile afresh, once
      mod = nullptr;
      rtc_cache_drop(src);
      code = rtc_compile(src, log, false);
      if (code.empty() || hipModuleLoadData(&mod, code.data()) != hipSuccess) {
        log += " hipModuleLoadData failed"; mod = nullptr; return false;
      }
    }
    if (hipModuleGetFunction(&fn, mod, entry) != hipSuccess) { log += " entry point not found"; return false; }
    ok = true;
    return true;
  }
  // does a workgroup of `waves` wavefronts of the loaded entry point fit a compute unit's register files?  (gfx950: 512
  // unified registers per lane and SIMD, four SIMDs: ceil(waves / 4) wavefronts share one.)  A launch that does not fit
  // aborts the queue (INVALID_ISA), so the callers that choose their own workgroup width ask first.
  bool fits(int waves) const {
    int regs = 0;
    if (!fn || hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, fn) != hipSuccess) return false;
    return regs * ((waves + 3) / 4) <= 512;
  }
  // further entry points of the same module (nullptr when missing)
  hipFunction_t get(const char* entry) {
    hipFunction_t f = nullptr;
    if (!mod || hipModuleGetFunction(&f, mod, entry) != hipSuccess) return nullptr;
    return f;
  }
};

}  // namespace dnlp
