/* placeholder until the register-systolic kernel lands: reports width 0 so AUTO picks the general kernel */
#include "cpecan_device.h"
extern "C" int cpecan_systolic_max_width(void) { return 0; }
extern "C" long long cpecan_systolic_ring_doubles(const DevParams *) { return 0; }
extern "C" int cpecan_systolic_workers(int, long long) { return 0; }
extern "C" int cpecan_systolic_launch(hipStream_t, const DevItem *, long long, DevParams,
                                      const long long *, const unsigned short *, const double *,
                                      const double *, double *, long long, int *, long long *,
                                      double *, long long *, long long *, double *, long long *,
                                      long long *, double *, int) { return -1; }
