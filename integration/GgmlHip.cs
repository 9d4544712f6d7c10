// GgmlHip.cs -- P/Invoke declarations for libggml_hip.so (include/ggml_hip.h), to be added to the GGMLSharp project
// (GGMLSharp/GgmlHip.cs).  The text of INTEGRATION.md section 1, as a file a maintainer can drop in; integration/README.md
// says how to build.  NOT compiled here: neither the build image nor the GPU boxes carry a .NET toolchain.
using System.Runtime.InteropServices;

namespace GGMLSharp;

internal static unsafe partial class GgmlHip
{
    const string Lib = "ggml_hip";   // libggml_hip.so on the loader path

    // include/ggml_hip.h -- status codes
    public const int OK = 0, ERR_NO_DEVICE = -1, ERR_TYPE = -2, ERR_SHAPE = -3, ERR_ARG = -4, ERR_RUNTIME = -5;

    [DllImport(Lib)] public static extern int ggml_hip_device_count();
    [DllImport(Lib)] public static extern int ggml_hip_init(int device);                          // one device slot
    [DllImport(Lib)] public static extern int ggml_hip_init_devices(int nDevices, int* deviceIds);   // n slots: Seam 1 row-splits over them (ids null: 0..n-1)
    [DllImport(Lib)] public static extern int ggml_hip_bind_thread(int slot);                     // this managed thread's seams run on one slot (-1: all)
    [DllImport(Lib)] public static extern void ggml_hip_shutdown();
    [DllImport(Lib)] public static extern sbyte* ggml_hip_last_error();
    // the context pool is ONE allocation (Ggml.cs:1545): register it and Seam 1 moves src1 / dst by asynchronous DMA in chunks
    [DllImport(Lib)] public static extern int ggml_hip_register_host_pool(void* pool, nuint bytes);
    [DllImport(Lib)] public static extern int ggml_hip_unregister_host_pool(void* pool);

    // Seam 1: drop-in for Ggml.ggml_compute_forward_mul_mat (Ggml.cs:6714-6744).
    // ggml_compute_params / ggml_tensor are blittable and laid out exactly as TypeDefinitions.cs:299-308 / 65-99.
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_mul_mat(
        ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* src1, ggml_tensor* dst);
    [DllImport(Lib)] public static extern void ggml_hip_invalidate(void* hostPtr);
    [DllImport(Lib)] public static extern void ggml_hip_invalidate_range(void* hostPtr, nuint bytes);
    [DllImport(Lib)] public static extern void ggml_hip_invalidate_all();
    [DllImport(Lib)] public static extern int ggml_hip_graph_begin();   // results of offloaded nodes stay in HBM for their consumers
    [DllImport(Lib)] public static extern int ggml_hip_graph_begin_keyed(ulong key);   // the same, NAMED: captured / replayed when it recurs
    [DllImport(Lib)] public static extern int ggml_hip_graph_end();     // ... until here; all node data is on the host afterwards
    [DllImport(Lib)] public static extern int ggml_hip_host_read(void* p, nuint bytes);   // a CPU node inside a scope reads an offloaded result
    [DllImport(Lib)] public static extern int ggml_hip_graph_outputs(void** ptrs, int n);  // OPT-IN, not the reference's contract: only these results go home
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_mul_mat_multi(ggml_compute_params* p, int n, ggml_tensor** src0,
        ggml_tensor* src1, ggml_tensor** dst, ggml_tensor* pro_x, ggml_tensor* pro_g, ggml_tensor* pro_norm);   // q / k / v, gate / up: one call

    // neighbours of the path (SURVEY 8(f)): same calling convention, dispatched from ggml_compute_forward (Ggml.cs:8548-8760)
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_cpy(ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* dst);           // f32/f16 -> Q
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_add(ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* src1, ggml_tensor* dst);   // Q + f32, f32 + f32
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_mul(ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* src1, ggml_tensor* dst);
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_scale(ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* src1, ggml_tensor* dst);
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_rms_norm(ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* dst);
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_silu(ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* dst);
    // fused pairs (SURVEY 8(f) row 4): node i and the node i + 1 that consumes it, one call, both nodes' data produced
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_rms_norm_mul(ggml_compute_params* @params, ggml_tensor* x, ggml_tensor* g, ggml_tensor* normDst, ggml_tensor* mulDst);
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_silu_mul(ggml_compute_params* @params, ggml_tensor* a, ggml_tensor* b, ggml_tensor* siluDst, ggml_tensor* mulDst);
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_mul_mat_add(ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* src1, ggml_tensor* mmDst, ggml_tensor* addend, ggml_tensor* addDst);
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_mul_mat_scale(ggml_compute_params* @params, ggml_tensor* src0, ggml_tensor* src1, ggml_tensor* mmDst, ggml_tensor* scalar, ggml_tensor* scaleDst);
    // rms_norm, mul, mul_mat [, add]: the pre-projection chain of a decoder block, ONE launch for decode-sized batches (addend / addDst null without an add node)
    [DllImport(Lib)] public static extern int ggml_hip_compute_forward_norm_mul_mat(ggml_compute_params* @params, ggml_tensor* x, ggml_tensor* g, ggml_tensor* normDst, ggml_tensor* mulDst, ggml_tensor* src0, ggml_tensor* mmDst, ggml_tensor* addend, ggml_tensor* addDst);

    // Seam 2: the quantize_fns_t slots (TypeDefinitions.cs:334-342), type-indexed
    [DllImport(Lib)] public static extern int ggml_hip_quantize_row(int type, float* x, void* y, int k);
    [DllImport(Lib)] public static extern int ggml_hip_dequantize_row(int type, void* x, float* y, int k);
    [DllImport(Lib)] public static extern int ggml_hip_vec_dot(int type, int n, float* s, void* vx, void* vy);
}
