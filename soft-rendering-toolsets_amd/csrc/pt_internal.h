// Calls between the translation units of libsrt_hip.so that are not part of the C ABI.
#ifndef SRT_PT_INTERNAL_H
#define SRT_PT_INTERNAL_H

#include "srt_pt.h"

namespace srt {

// After the caller has synchronised the stream(s) the context rendered on: did a streamed launch end with unfinished work units
// (pt_stream_finish_kernel)?  SRT_ERR_STATE once, then the flag is lowered (what srt_pt_render_epoch / srt_pt_sync report themselves).
int pt_check_fault(srt_pt* pt, const char* what);

}  // namespace srt

#endif
