// oracle/shim/kilo_prefix.h — force-included (-include) in front of the reference's KILO.cc ONLY.
// KILO.cc:369 orders the points of a scan by time with the UNSTABLE std::sort, so the reference's own result is only
// defined up to a permutation inside each time bucket (the permutation decides the summation order of the update and
// which points of a voxel meet which refit).  Every stable order is one of the outcomes std::sort may produce; the macro
// below selects it, which is also the order the oracle and the device path define for themselves.
// All standard headers that mention the identifier are included first so that only KILO.cc's own call is rewritten.
#include <algorithm>
#include <chrono>
#include <deque>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>
#include <cmath>
#include <complex>
// KILO.cc's comparator takes non-const references (KILO.cc:19), which std::stable_sort's internal lower/upper_bound cannot
// bind; the adaptor below forwards through a const-correct lambda.  (It has to live in namespace std because the call
// site is spelled std::sort.)
namespace std {
template <class It, class Cmp>
void lk_stable_sort(It first, It last, Cmp cmp) {
    typedef typename std::iterator_traits<It>::value_type V;
    std::stable_sort(first, last, [&](const V& a, const V& b) { return cmp(const_cast<V&>(a), const_cast<V&>(b)); });
}
}  // namespace std
#define sort lk_stable_sort
