// oracle/shim — TEST INFRASTRUCTURE ONLY: the three boost::filesystem names common/glog_utils.hpp mentions
#ifndef LK_SHIM_BOOST_FS
#define LK_SHIM_BOOST_FS
#include <filesystem>
#include <string>
namespace boost {
namespace filesystem {
typedef std::filesystem::filesystem_error filesystem_error;
inline bool exists(const std::string& p) { return std::filesystem::exists(p); }
inline bool create_directory(const std::string& p) { return std::filesystem::create_directory(p); }
}  // namespace filesystem
}  // namespace boost
#endif
