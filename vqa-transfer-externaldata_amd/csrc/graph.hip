// Whole-step hipGraph capture / replay for callers of the whole-model entry points.
//
// vqa_fusion_forward / _backward, vqa_pretrain_forward / _backward, vqa_sumsq and vqa_clip_adam_dev only ENQUEUE work
// (no allocation, no synchronisation, no host-to-device copy: include/vqa_hot.h), the recurrence's row chains fork and
// join their side streams with events, so a caller can put a stream into capture, issue one whole train step and get a
// replayable executable graph back: every later step is ONE host call instead of ~130 launches.  What a replay cannot
// change are the kernel arguments baked in at capture -- pointers must stay valid and step-dependent scalars must be
// read from device memory (vqa_clip_adam_dev takes lr_t from a device float for that reason).
// Relaxed capture mode: other threads of the process (a framework's allocator, a data loader) may keep calling the
// runtime while this thread captures.
#include "vqa_common.h"

extern "C" int vqa_graph_capture_begin(void* stream) {
    VQA_REQUIRE(stream != nullptr, VQA_ERR_ARG);        // the NULL stream cannot be captured
    return hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeRelaxed) == hipSuccess ? VQA_OK
                                                                                                             : VQA_ERR_LAUNCH;
}

extern "C" int vqa_graph_capture_end(void* stream, void** exec_out, int* n_nodes_out) {
    VQA_REQUIRE(stream != nullptr && exec_out != nullptr, VQA_ERR_ARG);
    *exec_out = nullptr;
    hipGraph_t graph = nullptr;
    if (hipStreamEndCapture(static_cast<hipStream_t>(stream), &graph) != hipSuccess || graph == nullptr) {
        (void)hipGetLastError();
        return VQA_ERR_LAUNCH;
    }
    if (n_nodes_out != nullptr) {
        size_t n = 0;
        *n_nodes_out = hipGraphGetNodes(graph, nullptr, &n) == hipSuccess ? (int)n : -1;
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess || exec == nullptr) return VQA_ERR_LAUNCH;
    *exec_out = exec;
    return VQA_OK;
}

// drops a capture that failed half-way (an entry point returned an error while the stream was capturing)
extern "C" int vqa_graph_capture_abort(void* stream) {
    VQA_REQUIRE(stream != nullptr, VQA_ERR_ARG);
    hipGraph_t graph = nullptr;
    (void)hipStreamEndCapture(static_cast<hipStream_t>(stream), &graph);
    if (graph != nullptr) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return VQA_OK;
}

extern "C" int vqa_graph_launch(void* exec, void* stream) {
    VQA_REQUIRE(exec != nullptr, VQA_ERR_ARG);
    return hipGraphLaunch(static_cast<hipGraphExec_t>(exec), static_cast<hipStream_t>(stream)) == hipSuccess ? VQA_OK
                                                                                                             : VQA_ERR_LAUNCH;
}

extern "C" int vqa_graph_destroy(void* exec) {
    if (exec == nullptr) return VQA_OK;
    return hipGraphExecDestroy(static_cast<hipGraphExec_t>(exec)) == hipSuccess ? VQA_OK : VQA_ERR_LAUNCH;
}

extern "C" int vqa_stream_is_capturing(void* stream) {
    hipStreamCaptureStatus s = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &s) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return s == hipStreamCaptureStatusActive ? 1 : 0;
}
