// oracle/ref_tum_capi.cc — TEST INFRASTRUCTURE ONLY.
// The REFERENCE's own TUM writer, legkilo::TrajectorySaver::write (common/trajectory_saver.hpp:43-50), compiled from
// /root/reference against oracle/shim (the quaternion conversion is Eigen's, restated in shim/Eigen/Dense).  The
// saver picks its own file name under ROOT_DIR "result/traj/"; the wrapper returns it.
#include <chrono>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>

#include <glog/logging.h>
#include <boost/filesystem.hpp>
#include "common/eigen_types.hpp"
// every header trajectory_saver.hpp includes is in by now: only the class itself sees the neutralised specifier
#define private public
#include "common/trajectory_saver.hpp"
#undef private

using namespace legkilo;

extern "C" int lkt_write_tum(const double* stamps, const double* rots9, const double* pos3, size_t n, char* path_out, size_t path_cap) {
    try {
        TrajectorySaver saver;
        for (size_t i = 0; i < n; ++i) {
            Mat3D R;
            Vec3D p;
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) R(r, c) = rots9[9 * i + 3 * r + c];
                p[r] = pos3[3 * i + r];
            }
            saver.write(stamps[i], R, p);
        }
        saver.flush();
        std::strncpy(path_out, saver.filepath_.c_str(), path_cap - 1);
        path_out[path_cap - 1] = 0;
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}
