// PinEmitter -- runs the UNTOUCHED reference (kant2002/GGMLSharp) and prints, as one JSON document, the inputs and outputs of
// the functions of the quantized mul_mat path that are correct as written (SURVEY.md defect ledger): the scalar
// quantize_row_q4_0_reference, the scalar dequantize_row_q4_0 / _q4_1 / _q5_0, and the f32 ggml_mul_mat on Test3's LCG data
// (Test3/Program.cs:33-42).  tests/test_reference_pins.py of the MI355X build compares its CPU oracle (and, on a GPU box, the
// HIP path) with this file bit for bit: it is what turns "parity unpinned" into a pinned oracle the day anyone has `dotnet`.
// Private members are reached through reflection: the reference is not modified.
using System.Reflection;
using System.Text;
using GGMLSharp;
using static GGMLSharp.Ggml;

unsafe
{
    ulong next = 1;
    int xrand() { next = next * 214013L + 2531011; return (int)((next >> 16) & 0x7FFF); }   // Test3/Program.cs:98-102
    float frand() => (float)xrand() / 32767.0f - 0.5f;

    static string Hex(byte* p, int n) { var sb = new StringBuilder(2 * n); for (int i = 0; i < n; i++) sb.Append(p[i].ToString("x2")); return sb.ToString(); }
    MethodInfo Priv(string name, params Type[] sig) =>
        typeof(Ggml).GetMethod(name, BindingFlags.NonPublic | BindingFlags.Public | BindingFlags.Static, null, sig, null)
        ?? throw new MissingMethodException("Ggml." + name);

    var cases = new List<string>();
    const int K = 256;                                             // eight blocks per row
    const int ROWS = 16;

    // ---- quantize_row_q4_0_reference(float* x, void* y, int k): rows with exact .5 ties, a zero block, a negative-max block ----
    {
        var q = Priv("quantize_row_q4_0_reference", typeof(float*), typeof(void*), typeof(int));
        float* x = stackalloc float[K];
        byte* y = stackalloc byte[K / 32 * 20];
        next = 0;
        for (int r = 0; r < ROWS; r++)
        {
            for (int i = 0; i < K; i++) x[i] = frand() * (r % 4 == 0 ? 16.0f : 1.0f);
            if (r == 1) for (int i = 0; i < 32; i++) x[i] = 0.0f;
            if (r == 2) { x[0] = -8.0f; x[1] = 0.5f; x[2] = 1.5f; x[3] = 2.5f; x[4] = -0.5f; x[5] = -1.5f; for (int i = 6; i < 32; i++) x[i] = 0.0f; }
            if (r == 3) for (int i = 0; i < 32; i++) x[i] = ((i % 16) - 8) * 0.5f;
            q.Invoke(null, new object[] { Pointer.Box(x, typeof(float*)), Pointer.Box(y, typeof(void*)), K });
            cases.Add($"{{\"fn\":\"quantize_row_q4_0_reference\",\"k\":{K},\"input\":\"{Hex((byte*)x, 4 * K)}\",\"output\":\"{Hex(y, K / 32 * 20)}\"}}");
        }
    }
    // ---- dequantize_row_q4_0 / q4_1 / q5_0 (void* vx, float* y, int k): every byte pattern of quants, finite scales ----
    foreach (var (name, bsz) in new[] { ("dequantize_row_q4_0", 20), ("dequantize_row_q4_1", 24), ("dequantize_row_q5_0", 22) })
    {
        var d = Priv(name, typeof(void*), typeof(float*), typeof(int));
        int nb = K / 32;
        byte* blk = stackalloc byte[nb * bsz];
        float* y = stackalloc float[K];
        next = (ulong)bsz;
        for (int r = 0; r < ROWS; r++)
        {
            for (int i = 0; i < nb * bsz; i++) blk[i] = (byte)(xrand() & 0xFF);
            for (int b = 0; b < nb; b++)
            {
                if (bsz == 22) { *(Half*)(blk + b * bsz) = (Half)(frand() * 0.25f); }                     // q5_0: half scale (read through an explicit Half -> float cast)
                else { *(float*)(blk + b * bsz) = frand() * 0.25f; if (bsz == 24) *(float*)(blk + b * bsz + 4) = frand(); }
            }
            d.Invoke(null, new object[] { Pointer.Box(blk, typeof(void*)), Pointer.Box(y, typeof(float*)), K });
            cases.Add($"{{\"fn\":\"{name}\",\"k\":{K},\"input\":\"{Hex(blk, nb * bsz)}\",\"output\":\"{Hex((byte*)y, 4 * K)}\"}}");
        }
    }
    // ---- f32 ggml_mul_mat through the public API, Test3's data recipe (BASELINE config 1: 64 x 128 x 256) ----
    {
        const int M = 64, KK = 128, N = 256;
        ggml_init_params ip = default;
        ip.mem_size = 64 * 1024 * 1024; ip.mem_buffer = null; ip.no_alloc = false;
        ggml_context* ctx = ggml_init(ip);
        ggml_tensor* W = ggml_new_tensor_2d(ctx, ggml_type.GGML_TYPE_F32, KK, M);
        ggml_tensor* X = ggml_new_tensor_2d(ctx, ggml_type.GGML_TYPE_F32, KK, N);
        next = 0;
        for (int i = 0; i < M * KK; i++) ((float*)W->data)[i] = frand() * 0.1f;
        for (int i = 0; i < N * KK; i++) ((float*)X->data)[i] = frand();
        ggml_tensor* Y = ggml_mul_mat(ctx, W, X);
        ggml_cgraph gf = ggml_build_forward(Y);
        gf.n_threads = 1;
        ggml_graph_compute(ctx, &gf);
        cases.Add($"{{\"fn\":\"mul_mat_f32\",\"M\":{M},\"K\":{KK},\"N\":{N},\"w\":\"{Hex((byte*)W->data, 4 * M * KK)}\",\"x\":\"{Hex((byte*)X->data, 4 * N * KK)}\",\"output\":\"{Hex((byte*)Y->data, 4 * M * N)}\"}}");
        ggml_free(ctx);
    }
    Console.Out.Write("{\"format\":\"ggmlsharp-reference-pins-v1\",\"runtime\":\"" + System.Runtime.InteropServices.RuntimeInformation.FrameworkDescription + "\",\"cases\":[\n" + string.Join(",\n", cases) + "\n]}\n");
}
