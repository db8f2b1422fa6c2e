// oracle/shim — TEST INFRASTRUCTURE ONLY: glog's LOG(...) as a sink that discards everything.
#ifndef LK_SHIM_GLOG
#define LK_SHIM_GLOG
#include <ios>
#include <ostream>
#include <string>
namespace lk_shim {
struct NullStream {
    template <class T>
    NullStream& operator<<(const T&) { return *this; }
    NullStream& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
    NullStream& operator<<(std::ios_base& (*)(std::ios_base&)) { return *this; }
};
}  // namespace lk_shim
namespace google {
enum { INFO = 0, WARNING = 1, ERROR = 2, FATAL = 3 };
inline void InitGoogleLogging(const char*) {}
inline void ShutdownGoogleLogging() {}
inline void FlushLogFiles(int) {}
inline void ParseCommandLineFlags(int*, char***, bool) {}
}  // namespace google
static int FLAGS_stderrthreshold = 0;
static bool FLAGS_colorlogtostderr = false;
static std::string FLAGS_log_dir;
#define LOG(severity) ::lk_shim::NullStream()
#define CHECK(cond) ::lk_shim::NullStream()
#endif
