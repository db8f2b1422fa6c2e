// oracle/shim — nothing of gflags is needed beyond what glog/logging.h (shim) declares
#include "../glog/logging.h"
