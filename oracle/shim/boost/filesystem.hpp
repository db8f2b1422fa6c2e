// oracle/shim — TEST INFRASTRUCTURE ONLY: the boost::filesystem names common/glog_utils.hpp and common/trajectory_saver.hpp mention
#ifndef LK_SHIM_BOOST_FS
#define LK_SHIM_BOOST_FS
#include <filesystem>
#include <string>
namespace boost {
namespace filesystem {
typedef std::filesystem::filesystem_error filesystem_error;
inline bool exists(const std::string& p) { return std::filesystem::exists(p); }
inline bool create_directory(const std::string& p) { return std::filesystem::create_directory(p); }
inline bool create_directories(const std::string& p) { return std::filesystem::create_directories(p); }
class path {
    std::filesystem::path p_;

   public:
    path() {}
    path(const std::string& s) : p_(s) {}
    path(const std::filesystem::path& s) : p_(s) {}
    path operator/(const std::string& o) const { return path(p_ / o); }
    path operator/(const path& o) const { return path(p_ / o.p_); }
    std::string string() const { return p_.string(); }
};
}  // namespace filesystem
}  // namespace boost
#endif
