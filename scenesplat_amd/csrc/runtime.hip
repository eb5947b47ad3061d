// Small runtime queries of the C-ABI that have no kernel behind them.
#include "common.h"
#include "../../include/scenesplat_hip.h"

// Capture status of a HIP stream: 0 = not capturing, 1 = capturing, 2 = capture invalidated (an illegal call was made while the
// stream captured: the capture can only be abandoned; ending it crashes inside the runtime on this stack).  < 0: the query failed.
extern "C" int ss_stream_capture_status(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing(stream, &st);
  if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
  return st == hipStreamCaptureStatusNone ? 0 : (st == hipStreamCaptureStatusActive ? 1 : 2);
}

// Identity of the capture a stream is recording into (unique per capture sequence in the process); 0 when the stream is not
// actively capturing or the query fails.  Lets host-side caches tell "written inside THIS capture" from "written in an earlier one".
extern "C" unsigned long long ss_stream_capture_id(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipError_t e = hipStreamGetCaptureInfo(stream, &st, &id);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0ULL; }
  // ids start at 0 in some runtimes: shift by one so that 0 can mean "none"
  return st == hipStreamCaptureStatusActive ? id + 1ULL : 0ULL;
}
