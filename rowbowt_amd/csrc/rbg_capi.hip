// rbg_capi.hip -- the C-ABI of include/rbg.h: owns the host copy of the flat index and its
// HBM replica, stages host batches, launches the kernels of k_search.hip / k_locate.hip / k_markers.hip / k_build.hip.
// There is deliberately no CPU compute path in this library.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <map>

#include "../../include/rbg.h"
#include "rbg_dev.h"
#include "rbg_host.hpp"


// The C-ABI of include/rbg.h, by concern (one translation unit: the parts share the scratch pools, the options and the handle's definition in an
// anonymous namespace, and the kernels' launchers are declared once in rbg_dev.h):
#include "capi/core.ipp"          // the handle, options, pools, arena
#include "capi/upload_runs.ipp"   // run-indexed layout: tables on the device, composition of the k-mer depths
#include "capi/load.ipp"          // upload() and its budget / layout rules; load / convert / build entry points; info
#include "capi/device_api.ipp"    // *_dev entry points
#include "capi/hostpath.ipp"      // host-pointer entry points, micro-batching
#include "capi/text.ipp"          // rbg_align_text
#include "capi/seeds.ipp"         // markers, marker seeds, greedy seeding
#include "capi/replicas.ipp"      // replicas in one process, counters, RCCL
