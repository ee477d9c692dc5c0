// Run-time compilation of generated kernels (fused_codegen.h, lbfgs_codegen.h): hiprtc -> code
// object for gfx950 -> hipModuleLoadData.  Code objects are cached on disk by the hash of their
// source and the hiprtc version ($DNLP_KERNEL_CACHE, else $XDG_CACHE_HOME/dnlp_kernel_cache, else
// /tmp/dnlp_kernel_cache-<uid>), so a problem structure is compiled once per machine.  The cache
// directory is used only when it is a real directory owned by this user with mode 0700 (code objects
// are executed: a directory somebody else could have planted is ignored and the kernel is compiled
// afresh); a cached object that no longer loads is deleted and recompiled.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

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

inline std::string rtc_cache_path(const std::string& src, const std::string& arch) {
  const std::string dir = rtc_trusted_cache_dir();
  if (dir.empty()) return "";
  int major = 0, minor = 0;
  hiprtcVersion(&major, &minor);
  char name[64];
  std::snprintf(name, sizeof name, "%016llx.hsaco",
                static_cast<unsigned long long>(rtc_hash(src + arch + "#hiprtc" + std::to_string(major) + "." +
                                                         std::to_string(minor))));
  return dir + "/" + name;
}

// Compile `src` for gfx950.  Returns the code object (empty on failure, `log` has the compiler text).
inline std::vector<char> rtc_compile(const std::string& src, std::string& log, bool use_cache = true) {
  const std::string arch = "--offload-arch=gfx950";
  const std::string path = use_cache ? rtc_cache_path(src, arch) : std::string();
  if (path.empty()) use_cache = false;
  std::vector<char> code;
  if (use_cache) {
    if (FILE* fp = std::fopen(path.c_str(), "rb")) {
      std::fseek(fp, 0, SEEK_END);
      const long n = std::ftell(fp);
      std::fseek(fp, 0, SEEK_SET);
      code.resize(static_cast<size_t>(n > 0 ? n : 0));
      const size_t got = code.empty() ? 0 : std::fread(code.data(), 1, code.size(), fp);
      std::fclose(fp);
      if (got == code.size() && !code.empty()) return code;
      code.clear();
    }
  }
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
  if (use_cache && !code.empty()) {
    // (unique per writer: the slots of a batch stream compile the same source on their own threads at the same time)
    static std::atomic<unsigned> writer{0};
    const std::string tmp = path + ".tmp" + std::to_string(static_cast<long>(getpid())) + "." + std::to_string(writer.fetch_add(1));
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
      const std::string path = rtc_cache_path(src, "--offload-arch=gfx950");
      if (!path.empty()) std::remove(path.c_str());
      code = rtc_compile(src, log, false);
      if (code.empty() || hipModuleLoadData(&mod, code.data()) != hipSuccess) {
        log += " hipModuleLoadData failed"; mod = nullptr; return false;
      }
    }
    if (hipModuleGetFunction(&fn, mod, entry) != hipSuccess) { log += " entry point not found"; return false; }
    ok = true;
    return true;
  }
  // further entry points of the same module (nullptr when missing)
  hipFunction_t get(const char* entry) {
    hipFunction_t f = nullptr;
    if (!mod || hipModuleGetFunction(&f, mod, entry) != hipSuccess) return nullptr;
    return f;
  }
};

}  // namespace dnlp
