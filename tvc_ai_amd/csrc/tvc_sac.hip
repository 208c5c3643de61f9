// libtvc_hip.so -- SAC learner: network definitions, forward/backward executor over the kernels of
// tvc_nn_kernels.h, the SAC loss/elementwise kernels, Adam/Polyak, and the C ABI (include/tvc_native.h).
//
// Replaces MultiAlgorithmAgent._create_sac_agent / get_action (policy part) / update / _update_sac /
// PhysicsInformedLoss, agent/multi_algorithm_agent.py:587-627, 736-809, 868-912, 950-1016, 236-285; further down the
// small-MLP handle behind CuriosityModule (env/enhanced_rocket_tvc_env.py:226-269), SafetyLayer (agent/...:287-351)
// and the goal policy of HierarchicalAgent (agent/...:353-417).
//
// Layout of this file: net description (Op / NetDef, build_actor / build_critic / derive_infer: the acting net with
// W_o W_v and the embedding folded) -> executor (net_forward / net_backward: which kernel runs each op, incl. the
// fused acting launches and the train-mode dropout sites) -> SAC elementwise kernels -> handle + C ABI.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "tvc_common.h"
#include "tvc_nn_kernels.h"
#include "tvc_actor_rows.h"
#include "tvc_actor_split.h"
#include "tvc_actor_x3.h"

using namespace tvcnn;

namespace {

// ------------------------------------------------------------------ network description
enum { OP_LINEAR = 0, OP_LN = 1, OP_HEAD = 2 };
struct Op {
    int type, in_dim, out_dim, act, src, res, rowtab;
    int ext = 0;  // 1: weight/bias offsets address the derived-weights buffer (folded W_o W_v of the acting net)
    int mul = -1;  // buffer multiplied into the output after the activation (SqueezeExcitation gate); thin layers only
    int drop = 0;  // train-mode dropout on this op's output: 0 none, 1 elementwise, g > 1 one mask per g columns (attention head)
    long w, b;  // offsets inside the net's parameter block (LINEAR/HEAD: weight[out,in], bias[out]; LN: gamma, beta)
};
struct TensorInfo { std::string name; long off; int rows, cols; };
struct NetDef {
    std::vector<Op> ops;
    std::vector<TensorInfo> tensors;
    long n_params = 0;
    int in_dim = 0;
    std::vector<int> buf_dim, res_consumer, producer, last_use;
    int add(int type, const std::string& name, int in, int out, int act, int src, int res, int rowtab) {
        Op o;
        o.type = type; o.in_dim = in; o.out_dim = out; o.act = act; o.src = src; o.res = res; o.rowtab = rowtab;
        o.w = n_params;
        if (type == OP_LN) {
            tensors.push_back({name + ".weight", n_params, out, 1}); n_params += out;
            o.b = n_params;
            tensors.push_back({name + ".bias", n_params, out, 1}); n_params += out;
        } else {
            tensors.push_back({name + ".weight", n_params, out, in}); n_params += (long)out * in;
            o.b = n_params;
            tensors.push_back({name + ".bias", n_params, out, 1}); n_params += out;
        }
        n_params = (n_params + 3) & ~3L;  // keep every tensor 16-byte aligned
        ops.push_back(o);
        return (int)ops.size();  // index of the output buffer
    }
    void finish() {
        const int nb = (int)ops.size() + 1;
        buf_dim.assign(nb, 0); res_consumer.assign(nb, -1); producer.assign(nb, -1); last_use.assign(nb, -1);
        buf_dim[0] = in_dim;
        for (int i = 0; i < (int)ops.size(); ++i) {
            buf_dim[i + 1] = ops[i].out_dim;
            producer[i + 1] = i;
            last_use[ops[i].src] = i;
            if (ops[i].res >= 0) { res_consumer[ops[i].res] = i; last_use[ops[i].res] = i; }
            if (ops[i].mul >= 0) last_use[ops[i].mul] = i;
        }
        last_use[nb - 1] = (int)ops.size();
    }
};

static NetDef build_actor(const tvc_sac_cfg& c) {
    NetDef n;
    n.in_dim = c.obs_dim;
    if (c.family == 0) {
        const int d = c.d_model;
        int x = n.add(OP_LINEAR, "input_embedding", c.obs_dim, d, ACT_NONE, 0, -1, 1);
        for (int l = 0; l < c.n_layers; ++l) {
            const std::string p = "layers." + std::to_string(l) + ".";
            // dropout sites of nn.TransformerEncoderLayer at sequence length 1: attention weights (softmax over one key = 1,
            // so the mask zeroes / rescales a whole head of V), dropout1, the FFN's dropout, dropout2
            int v = n.add(OP_LINEAR, p + "v_proj", d, d, ACT_NONE, x, -1, 0);
            n.ops.back().drop = std::max(1, d / std::max(1, (int)c.nhead));
            int a = n.add(OP_LINEAR, p + "out_proj", d, d, ACT_NONE, v, x, 0);
            n.ops.back().drop = 1;
            int x1 = n.add(OP_LN, p + "norm1", d, d, 0, a, -1, 0);
            int f = n.add(OP_LINEAR, p + "linear1", d, c.ff_dim, ACT_GELU, x1, -1, 0);
            n.ops.back().drop = 1;
            int g = n.add(OP_LINEAR, p + "linear2", c.ff_dim, d, ACT_NONE, f, x1, 0);
            n.ops.back().drop = 1;
            x = n.add(OP_LN, p + "norm2", d, d, 0, g, -1, 0);
        }
        x = n.add(OP_LN, "feature_norm", d, d, 0, x, -1, 0);
        if (c.use_se) {  // SqueezeExcitation(d, reduction 16) of agent/...:104-118: x * sigmoid(fc2(relu(fc1(x)))); inference only
            const int y = n.add(OP_LINEAR, "se_block.fc1", d, d / 16, ACT_RELU, x, -1, 0);
            const int g = n.add(OP_LINEAR, "se_block.fc2", d / 16, d, ACT_SIGMOID, y, -1, 0);
            n.ops.back().mul = x;
            x = g;
        }
        x = n.add(OP_LINEAR, "policy_head.0", d, c.head1, ACT_GELU, x, -1, 0);
        x = n.add(OP_LN, "policy_head.2", c.head1, c.head1, 0, x, -1, 0);
        n.ops.back().drop = 1;
        x = n.add(OP_LINEAR, "policy_head.4", c.head1, c.head2, ACT_GELU, x, -1, 0);
        x = n.add(OP_LN, "policy_head.6", c.head2, c.head2, 0, x, -1, 0);
        n.ops.back().drop = 1;
        n.add(OP_HEAD, "policy_head.8", c.head2, 2 * c.act_dim, 0, x, -1, 0);
    } else {
        int x = n.add(OP_LINEAR, "0", c.obs_dim, c.mlp1, ACT_RELU, 0, -1, 0);
        x = n.add(OP_LINEAR, "2", c.mlp1, c.mlp2, ACT_RELU, x, -1, 0);
        n.add(OP_HEAD, "4", c.mlp2, 2 * c.act_dim, 0, x, -1, 0);
    }
    n.finish();
    return n;
}
// Acting-only variant of the actor: at sequence length 1 attention is out_proj(v_proj(x)), two
// back-to-back Linear layers with nothing in between, so for inference they fold into ONE 256x256 Linear with
// W_ov = W_o W_v, b_ov = W_o b_v + b_o (re-derived after every actor update; the training net keeps them apart).
// ... and, when every row uses PE(0) (pe_rows == 1) and the observation is at most 16 wide, the embedding folds into
// the first of them: LN(x0 + W_ov x0 + b_ov) with x0 = W_e obs + b_e + pe0 is LN(W' obs + b'), one thin Linear + LN.
struct FoldInfo {
    int layers = 0; long v_w = 0, v_b = 0, o_w = 0, o_b = 0, layer_stride = 0; int d = 0;
    bool embed = false; long e_w = 0, e_b = 0, e_off = 0; int obs = 0;  // embedding fold: source offsets, offset in the derived buffer
};
static NetDef derive_infer(const NetDef& a, FoldInfo& fi, bool fold_embed) {
    NetDef n;
    n.in_dim = a.in_dim;
    std::vector<int> map(a.ops.size() + 1, -1);
    map[0] = 0;
    for (size_t i = 0; i < a.ops.size(); ++i) {
        const Op& op = a.ops[i];
        const bool fold = op.type == OP_LINEAR && op.act == ACT_NONE && op.res < 0 && op.in_dim == op.out_dim && !op.rowtab &&
                          i + 1 < a.ops.size() && a.ops[i + 1].type == OP_LINEAR && a.ops[i + 1].act == ACT_NONE &&
                          a.ops[i + 1].src == (int)i + 1 && a.ops[i + 1].res == op.src && a.ops[i + 1].in_dim == op.out_dim;
        if (fold) {
            const Op& o2 = a.ops[i + 1];
            if (fi.layers == 0) { fi.v_w = op.w; fi.v_b = op.b; fi.o_w = o2.w; fi.o_b = o2.b; fi.d = op.in_dim; }
            if (fi.layers == 1) fi.layer_stride = op.w - fi.v_w;
            Op f = o2;
            f.src = map[op.src]; f.res = map[op.src]; f.ext = 1;
            f.w = (long)fi.layers * ((long)fi.d * fi.d + fi.d); f.b = f.w + (long)fi.d * fi.d;
            n.ops.push_back(f);
            map[i + 2] = (int)n.ops.size();
            fi.layers += 1;
            ++i;
        } else {
            Op c = op;
            c.src = map[op.src];
            c.res = op.res >= 0 ? map[op.res] : -1;
            c.mul = op.mul >= 0 ? map[op.mul] : -1;
            n.ops.push_back(c);
            map[i + 1] = (int)n.ops.size();
        }
    }
    // ops[0] = embedding (src 0, row table), ops[1] = folded attention of layer 0 reading and adding buffer 1, ops[2] = norm1
    if (fold_embed && fi.layers > 0 && n.ops.size() >= 3 && n.ops[0].type == OP_LINEAR && n.ops[0].rowtab && n.ops[0].src == 0 &&
        n.ops[0].in_dim <= THIN_K && n.ops[1].ext == 1 && n.ops[1].src == 1 && n.ops[1].res == 1 && n.ops[2].type == OP_LN &&
        n.ops[2].src == 2) {
        bool only = true;  // buffers 1 and 2 must have no other reader
        for (size_t i = 2; i < n.ops.size(); ++i) {
            const Op& o = n.ops[i];
            if (o.src == 1 || o.res == 1 || o.mul == 1 || (i > 2 && (o.src == 2 || o.res == 2 || o.mul == 2))) only = false;
        }
        if (only) {
            fi.embed = true; fi.e_w = n.ops[0].w; fi.e_b = n.ops[0].b; fi.obs = n.ops[0].in_dim;
            fi.e_off = (((long)fi.layers * ((long)fi.d * fi.d + fi.d)) + 3) & ~3L;
            Op e = n.ops[0];
            e.rowtab = 0; e.ext = 1; e.w = fi.e_off; e.b = fi.e_off + (long)fi.d * fi.obs;
            std::vector<Op> ops;
            ops.push_back(e);
            for (size_t i = 2; i < n.ops.size(); ++i) {
                Op o = n.ops[i];
                auto shift = [](int b) { return b >= 2 ? b - 1 : b; };
                o.src = shift(o.src); o.res = o.res >= 0 ? shift(o.res) : -1; o.mul = o.mul >= 0 ? shift(o.mul) : -1;
                ops.push_back(o);
            }
            n.ops = ops;
        }
    }
    n.finish();
    return n;
}

static NetDef build_critic(const tvc_sac_cfg& c) {
    NetDef n;
    n.in_dim = c.obs_dim + c.act_dim;
    if (c.family == 0) {
        int x = n.add(OP_LINEAR, "0", n.in_dim, c.critic1, ACT_GELU, 0, -1, 0);
        x = n.add(OP_LN, "2", c.critic1, c.critic1, 0, x, -1, 0);
        n.ops.back().drop = 1;
        x = n.add(OP_LINEAR, "4", c.critic1, c.critic2, ACT_GELU, x, -1, 0);
        x = n.add(OP_LN, "6", c.critic2, c.critic2, 0, x, -1, 0);
        n.ops.back().drop = 1;
        n.add(OP_HEAD, "8", c.critic2, 1, 0, x, -1, 0);
    } else {
        int x = n.add(OP_LINEAR, "0", n.in_dim, c.critic1, ACT_RELU, 0, -1, 0);
        x = n.add(OP_LINEAR, "2", c.critic1, c.critic2, ACT_RELU, x, -1, 0);
        n.add(OP_HEAD, "4", c.critic2, 1, 0, x, -1, 0);
    }
    n.finish();
    return n;
}

static int validate(const tvc_sac_cfg* c) {
    if (!c) return tvc::set_error(TVC_EINVAL, "cfg is NULL");
    if (c->obs_dim < 1 || c->act_dim < 1 || c->act_dim > 2) return tvc::set_error(TVC_EINVAL, "obs_dim >= 1, act_dim in {1,2}");
    if (c->family != 0 && c->family != 1) return tvc::set_error(TVC_EINVAL, "family must be 0 or 1");
    auto ok_ln = [](int d) { return d == 256 || d == 512; };
    auto ok_k = [](int d) { return d >= 4 && (d % 4) == 0; };
    if (c->family == 0) {
        if (!ok_ln(c->d_model) || !ok_ln(c->head1) || !ok_ln(c->head2) || !ok_ln(c->critic1) || !ok_ln(c->critic2))
            return tvc::set_error(TVC_EINVAL, "family 0: LayerNorm widths (d_model, head1, head2, critic1, critic2) must be 256 or 512");
        if (!ok_k(c->ff_dim) || c->n_layers < 1 || c->n_layers > 16) return tvc::set_error(TVC_EINVAL, "bad ff_dim / n_layers");
    } else {
        if (!ok_k(c->mlp1) || !ok_k(c->mlp2) || !ok_k(c->critic1) || !ok_k(c->critic2))
            return tvc::set_error(TVC_EINVAL, "family 1: hidden widths must be multiples of 4");
    }
    if (c->batch_size < 1 || c->max_act_rows < 1) return tvc::set_error(TVC_EINVAL, "batch_size / max_act_rows must be >= 1");
    if (c->pe_rows < 1) return tvc::set_error(TVC_EINVAL, "pe_rows must be >= 1");
    if (c->use_se != 0 && c->use_se != 1) return tvc::set_error(TVC_EINVAL, "use_se must be 0 or 1");
    if (!(c->dropout_p >= 0.0f && c->dropout_p < 0.9f)) return tvc::set_error(TVC_EINVAL, "dropout_p must be in [0, 0.9)");
    if (c->dropout_p > 0.0f && c->family != 0) return tvc::set_error(TVC_EINVAL, "dropout_p needs family 0 (the MLP family has no dropout)");
    if (c->family == 0 && (c->nhead < 1 || (c->d_model % c->nhead) != 0)) return tvc::set_error(TVC_EINVAL, "nhead must divide d_model");
    if (c->use_se && (c->family != 0 || c->d_model != 256)) return tvc::set_error(TVC_EINVAL, "use_se needs family 0, d_model 256");
    return 0;
}

// ------------------------------------------------------------------ execution contexts
struct Ctx {
    int M = 0, G = 1;
    std::vector<float*> Y, dY, Z, mean, rstd;  // per buffer index
    std::vector<float*> dYm;  // masked copy of dY for a Linear whose dropped output fed a residual add (dropout on)
    std::vector<long> gY;                       // group stride (elements) per buffer
    unsigned* cnt = nullptr;                    // arrival counters of the Linear+LayerNorm launches (ln_tail), training contexts only
};
constexpr int kLnTailCounters = 256;

static int fuse_ln_min_rows() {  // acting batches at least this large use the fused Linear+LayerNorm kernel
    static const int v = [] { const char* e = getenv("TVC_FUSE_LN_MIN_ROWS"); return e ? atoi(e) : 6144; }();
    return v;
}
static int g_force_variant = 0;  // diagnostics: 0 auto, 1 = 64x64 LDS-tiled, 3 = skinny split-K

// FAST instantiations of the split-K kernel: full tiles, K in chunks of 256 (4 k-steps per wave), float4 everywhere
static bool skinny_fast_ok(const GemmArgs& g) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return (g.M % 32) == 0 && (g.N % 32) == 0 && g.K >= 256 && (g.K % 256) == 0 && ((g.lda | g.ldb | g.ldc) & 3) == 0 &&
           ((g.gA | g.gB | g.gC | g.gBias | g.gZ | g.gR | g.gDZ) & 3) == 0 && al(g.A) && al(g.B) && al(g.C) && al(g.bias) &&
           al(g.Zout) && al(g.Radd) && al(g.dactZ) && al(g.rowtab) && g.A2 == nullptr;
}
static void launch_gemm(bool a_kc, bool b_kc, const GemmArgs& g, int G, hipStream_t st) {
    const bool aligned = ((g.lda & 3) == 0) && ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
    const int fv = g_force_variant;
    const bool dropping = g.drop.ctr != nullptr || g.dmask.ctr != nullptr;  // only the split-K kernel's epilogue knows the masks
    if ((fv == 3 || dropping || (fv == 0 && (long)g.M * g.N <= 512L * 512L)) && g.A2 == nullptr) {
        // update path (batch of a few hundred rows): latency-optimised split-K kernel
        dim3 grid((g.N + 31) / 32, (g.M + 31) / 32, G), block(256);
        const bool fast = aligned && skinny_fast_ok(g);
#define TVC_SKINNY(AK, BK)                                                                               \
    do {                                                                                                 \
        if (fast) hipLaunchKernelGGL((gemm_skinny_kernel<AK, BK, true>), grid, block, 0, st, g);         \
        else hipLaunchKernelGGL((gemm_skinny_kernel<AK, BK, false>), grid, block, 0, st, g);             \
    } while (0)
        if (a_kc && b_kc) TVC_SKINNY(true, true);
        else if (a_kc && !b_kc) TVC_SKINNY(true, false);
        else if (!a_kc && b_kc) TVC_SKINNY(false, true);
        else TVC_SKINNY(false, false);
#undef TVC_SKINNY
        return;
    }
    dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, G), block(256);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, st, g);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, st, g);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, st, g);
}

// Linear whose A operand is LayerNorm(A rows) (one or two norms + dropout), computed in the operand load: update path
static void launch_gemm_ln(const GemmArgs& g, int G, hipStream_t st) {
    dim3 grid(g.N / 32, g.M / 32, G), block(256);
    if (g.K == 256) hipLaunchKernelGGL((gemm_skinny_ln_kernel<4>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_skinny_ln_kernel<8>), grid, block, 0, st, g);
}
// Linear + the LayerNorm(s) / head behind it in one launch (the last column tile to arrive normalises the rows): update path
static void launch_gemm_lnt(const GemmArgs& g, const LnTail& t, int G, hipStream_t st) {
    dim3 grid(g.N / 32, g.M / 32, G), block(256);
    if (g.N == 256) hipLaunchKernelGGL((gemm_skinny_lnt_kernel<4>), grid, block, 0, st, g, t);
    else hipLaunchKernelGGL((gemm_skinny_lnt_kernel<8>), grid, block, 0, st, g, t);
}
static int ln_tail_enabled() {
    static const int v = [] { const char* e = getenv("TVC_LN_TAIL"); return e ? atoi(e) : 0; }();
    return v;
}
// TVC_FOLD_LN=1 computes the update's LayerNorms inside the consumer GEMM's operand load (13 launches fewer, 92 -> 79).  OFF by
// default: measured SLOWER, 559 vs 535 us per update alone and 0.74 vs 0.71 ms per step at 4 096 envs (profiles/r03_c_update.md):
// a LayerNorm launch is cheap (4 KB of rows per workgroup), while the folded GEMM cannot start its MFMAs before all of its rows
// have arrived and been reduced, and its 186 - 256 VGPRs keep it from running beside the acting kernel.
static int fold_ln_enabled() {
    static const int v = [] { const char* e = getenv("TVC_FOLD_LN"); return e ? atoi(e) : 0; }();
    return v;
}

// fused Linear (+act, +residual) + LayerNorm(s), 32 complete rows per workgroup; the k-tile count is a template
// parameter for the reference shapes (K = 256, 512), which also gives each shape its own kernel name in a profile
static void launch_rowln(const RowLnArgs& a, hipStream_t st) {
    const dim3 grid((a.M + 31) / 32), block(256);
    const int ks = a.K / GBK;
    if (a.genX) {  // generated A operand (thin input layer folded in): K = the hidden width, at most 256
        if (a.N == 256) hipLaunchKernelGGL((gemm_rowln_kernel<4, 0, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((gemm_rowln_kernel<8, 0, true>), grid, block, 0, st, a);
        return;
    }
#define TVC_ROWLN(JT)                                                                               \
    do {                                                                                            \
        if (ks == 16) hipLaunchKernelGGL((gemm_rowln_kernel<JT, 16>), grid, block, 0, st, a);       \
        else if (ks == 32) hipLaunchKernelGGL((gemm_rowln_kernel<JT, 32>), grid, block, 0, st, a);  \
        else hipLaunchKernelGGL((gemm_rowln_kernel<JT, 0>), grid, block, 0, st, a);                 \
    } while (0)
    if (a.N == 256) TVC_ROWLN(4);
    else TVC_ROWLN(8);
#undef TVC_ROWLN
}

// dW = dZ^T X and dX = dZ W of one Linear in one launch when both fit the split-K kernel's fast path
static void launch_gemm_bwd_pair(const GemmArgs& w, const GemmArgs& x, int G, hipStream_t st) {
    auto fast = [](const GemmArgs& g) {
        return skinny_fast_ok(g) && ((long)g.M * g.N <= 512L * 512L || g.drop.ctr != nullptr || g.dmask.ctr != nullptr);
    };
    if (g_force_variant != 0 || !fast(w) || !fast(x)) {
        launch_gemm(false, false, w, G, st);
        launch_gemm(true, false, x, G, st);
        return;
    }
    GemmPair p;
    p.w = w; p.x = x;
    p.w_tiles_x = w.N / 32; p.x_tiles_x = x.N / 32;
    p.nw = p.w_tiles_x * (w.M / 32);
    const int nx = p.x_tiles_x * (x.M / 32);
    hipLaunchKernelGGL((gemm_skinny_bwd_kernel<true>), dim3(p.nw + nx, 1, G), dim3(256), 0, st, p);
}

// optional second input source of a net whose first layer reads a concatenation [X[:, :K1] | X2[:, :in_dim-K1]]
struct In2 { const float* X2; long gX2; int K1, ldx, ldx2; int gdiv = 1; };  // gdiv > 1: input group = z / gdiv
static ThinArgs thin_input(const Op& o, const float* X, long gX, const In2* in2, int M) {
    ThinArgs a{};
    if (o.src != 0) in2 = nullptr;  // only the net input can be a two-source concatenation
    a.X = X; a.gX = gX; a.K = o.in_dim; a.M = M; a.N = o.out_dim;
    a.K1 = in2 ? in2->K1 : o.in_dim; a.ldx = in2 ? in2->ldx : o.in_dim;
    a.X2 = in2 ? in2->X2 : nullptr; a.ldx2 = in2 ? in2->ldx2 : 0; a.gX2 = in2 ? in2->gX2 : 0;
    a.gdiv = in2 ? in2->gdiv : 1;
    return a;
}
static int thin_wgrad_slices(int M) {  // TVC_WGRAD_SLICES: batch slices of the thin layers' weight gradient (1 = no atomics)
    static const int v = [] { const char* e = getenv("TVC_WGRAD_SLICES"); return e ? atoi(e) : 4; }();
    return std::max(1, std::min(v, (M + 63) / 64));
}
static bool thin_ok(const Op& o) { return o.type == OP_LINEAR && o.in_dim <= THIN_K && o.res < 0; }

// train-mode dropout of one forward call (and of the backward that follows it): the counter, p, and the site base that
// tells this call's masks from every other call's
struct DropCtl { const int* ctr; unsigned thresh; float scale; unsigned site_base; unsigned seed; unsigned zsplit = 0, site_base2 = 0; };
static DropArgs drop_args(const DropCtl* dc, int op_index, int group) {
    DropArgs d{};
    if (dc && group > 0) {
        d.ctr = dc->ctr; d.site = dc->site_base + (unsigned)op_index; d.thresh = dc->thresh; d.scale = dc->scale; d.group = group;
        d.seed = dc->seed;
        d.zsplit = dc->zsplit; d.site2 = dc->site_base2 + (unsigned)op_index;
    }
    return d;
}

// LayerNorm ops starting at op i that can be computed inside the operand load of the Linear behind them (0 = none, 1, or 2 for
// norm2 -> feature_norm -> Linear): the consumer must take the split-K kernel's fast path and the norm must span its whole K
static int ln_fold_len(const NetDef& nd, int i, int M, bool dropping) {
    if (g_force_variant != 0 || !fold_ln_enabled()) return 0;
    const int nops = (int)nd.ops.size();
    const Op& o = nd.ops[i];
    if (o.type != OP_LN || (o.out_dim != 256 && o.out_dim != 512)) return 0;
    int n = 1;
    if (i + 1 < nops && nd.ops[i + 1].type == OP_LN && nd.ops[i + 1].src == i + 1 && nd.ops[i + 1].out_dim == o.out_dim && !o.drop) n = 2;
    const int li = i + n;
    if (li >= nops) return 0;
    const Op& l = nd.ops[li];
    if (l.type != OP_LINEAR || l.src != li || l.in_dim != o.out_dim || l.rowtab || l.mul >= 0) return 0;
    // the consumer may not ALSO read a folded norm's output as its residual: that buffer is written by column tile 0 of the same
    // launch (the acting net's folded attention x + W_ov x reads x twice; the training net's residual readers are later launches)
    if (l.res >= 0 && l.res > i && l.res <= li) return 0;
    if ((M % 32) != 0 || (l.out_dim % 32) != 0) return 0;
    if (!dropping && (long)M * l.out_dim > 512L * 512L) return 0;  // larger products go to the LDS-tiled kernel
    return n;
}

// forward of one net (G parameter groups batched through blockIdx.z).  X: [G?][M,in]; gX = 0 shares one input.
// LayerNorm ops (at most two) directly behind Linear op i, and the output head behind the last of them, as the tail of the
// Linear's own launch (tvc_nn_kernels.h: ln_tail); returns how many norms ride along (0: none / not available)
static int make_ln_tail(const NetDef& nd, int i, const float* P, long gP, int M, Ctx& c, bool save, const DropCtl* dc, LnTail& t,
                        bool& head) {
    const int nops = (int)nd.ops.size(), out = i + 1, width = nd.ops[i].out_dim;
    head = false;
    if (!c.cnt || !ln_tail_enabled() || g_force_variant != 0 || (width != 256 && width != 512)) return 0;
    int nln = 0;
    while (nln < 2 && i + 1 + nln < nops && nd.ops[i + 1 + nln].type == OP_LN && nd.ops[i + 1 + nln].src == out + nln &&
           nd.ops[i + 1 + nln].out_dim == width)
        ++nln;
    if (nln == 0) return 0;
    t = LnTail{};
    t.cnt = c.cnt;
    for (int k = 0; k < nln; ++k) {
        const int li = i + 1 + k, lout = li + 1;
        const Op& l = nd.ops[li];
        LnArgs& a = k == 0 ? t.n1 : t.n2;
        a.X = c.Y[l.src]; a.Y = c.Y[lout]; a.gamma = P + l.w; a.beta = P + l.b;
        a.mean = save ? c.mean[lout] : nullptr; a.rstd = save ? c.rstd[lout] : nullptr;
        a.M = M; a.N = l.out_dim; a.gX = c.gY[l.src]; a.gY = c.gY[lout]; a.gP = gP; a.gS = M;
        a.drop = drop_args(dc, li, l.drop);
    }
    const int last = i + nln, lastout = last + 1;  // op index / buffer index of the last norm
    if (last + 1 < nops && nd.ops[last + 1].type == OP_HEAD && nd.ops[last + 1].src == lastout && nd.ops[last + 1].out_dim <= 4) {
        const Op& ho = nd.ops[last + 1];
        LnArgs& a = nln == 1 ? t.n1 : t.n2;
        a.headW = P + ho.w; a.headB = P + ho.b; a.headOut = c.Y[lastout + 1]; a.headN = ho.out_dim;
        a.gHW = gP; a.gHO = c.gY[lastout + 1];
        head = true;
    }
    return nln;
}

static void net_forward(const NetDef& nd, const float* P, long gP, const float* X, long gX, int M, int G, Ctx& c, bool save,
                        const float* pe, int pe_rows, hipStream_t st, const float* Pext = nullptr, const In2* in2 = nullptr,
                        const DropCtl* dc = nullptr) {
    int fold_i = -1, fold_n = 0;  // LayerNorm ops waiting to be computed inside the next Linear's operand load
    for (int i = 0; i < (int)nd.ops.size(); ++i) {
        const Op& o = nd.ops[i];
        const int out = i + 1;
        const float* in = o.src == 0 ? X : c.Y[o.src];
        const long gin = o.src == 0 ? gX : c.gY[o.src];
        if (o.type == OP_LN && fold_n == 0) {
            const int n = ln_fold_len(nd, i, M, dc != nullptr);
            if (n > 0) { fold_i = i; fold_n = n; i += n - 1; continue; }
        }
        if (thin_ok(o) && g_force_variant == 0) {  // input layer (in_dim <= 16): plain-FMA kernel, reads [X | X2] in place
            ThinArgs a = thin_input(o, in, gin, in2, M);
            a.W = (o.ext ? Pext : P) + o.w; a.bias = (o.ext ? Pext : P) + o.b; a.gW = gP; a.gB = gP;
            if (o.rowtab && pe) { a.rowtab = pe; a.rowtab_rows = pe_rows; }
            if (o.mul >= 0) a.Mul = c.Y[o.mul];
            a.Y = c.Y[out]; a.gY = c.gY[out]; a.act = o.act;
            a.Z = (save && o.act != ACT_NONE) ? c.Z[out] : nullptr;
            // MLP family acting: input layer -> hidden Linear (+act) -> output head as ONE launch; the input layer is
            // evaluated inside the hidden layer's k-loop
            if (!save && !dc && G == 1 && M >= fuse_ln_min_rows() && o.src == 0 && !in2 && !o.rowtab && o.mul < 0 && o.out_dim <= 256 &&
                (o.out_dim % GBK) == 0 && i + 2 < (int)nd.ops.size() && nd.last_use[out] == i + 1) {
                const Op& h = nd.ops[i + 1];
                const Op& ho = nd.ops[i + 2];
                if (h.type == OP_LINEAR && h.src == out && h.res < 0 && !h.rowtab && !h.ext && (h.out_dim == 256 || h.out_dim == 512) &&
                    ho.type == OP_HEAD && ho.src == out + 1 && nd.last_use[out + 1] == i + 2 && ho.out_dim <= 4) {
                    RowLnArgs r{};
                    r.B = P + h.w; r.M = M; r.N = h.out_dim; r.K = h.in_dim; r.ldb = h.in_dim; r.ldc = h.out_dim;
                    r.bias = P + h.b; r.act = h.act;
                    r.headW = P + ho.w; r.headB = P + ho.b; r.headOut = c.Y[out + 2]; r.headN = ho.out_dim;
                    r.genX = X; r.genW = (o.ext ? Pext : P) + o.w; r.genB = (o.ext ? Pext : P) + o.b; r.genK = o.in_dim;
                    r.genAct = o.act; r.genLd = o.in_dim;
                    launch_rowln(r, st);
                    i += 2;
                    continue;
                }
            }
            if (!save && !dc && G == 1 && o.out_dim == 256 && o.mul < 0 && !(o.rowtab && pe) && i + 1 < (int)nd.ops.size() &&
                nd.ops[i + 1].type == OP_LN && nd.ops[i + 1].src == out && nd.last_use[out] == i + 1) {
                const Op& ln = nd.ops[i + 1];  // thin Linear + LayerNorm in one launch (the folded embedding block)
                a.Y = c.Y[out + 1];
                hipLaunchKernelGGL(thin_fwd_ln_kernel, dim3((M + THIN_ROWS - 1) / THIN_ROWS), dim3(256), 0, st, a, P + ln.w, P + ln.b);
                i += 1;
                continue;
            }
            hipLaunchKernelGGL(thin_fwd_kernel, dim3((M + THIN_ROWS - 1) / THIN_ROWS, (o.out_dim + 255) / 256, G), dim3(256), 0, st, a);
            continue;
        }
        // acting pass (nothing saved, many rows): Linear (+act, +residual) and the LayerNorm(s) behind it in ONE launch,
        // 32 complete rows per workgroup
        if (o.type == OP_LINEAR && !save && !dc && G == 1 && M >= fuse_ln_min_rows() && !o.rowtab && (o.in_dim % GBK) == 0 &&
            (o.out_dim == 256 || o.out_dim == 512) && i + 1 < (int)nd.ops.size() && nd.ops[i + 1].type == OP_LN &&
            nd.ops[i + 1].src == out && nd.last_use[out] == i + 1 && g_force_variant == 0) {
            const Op& ln = nd.ops[i + 1];
            const bool two = i + 2 < (int)nd.ops.size() && nd.ops[i + 2].type == OP_LN && nd.ops[i + 2].src == out + 1 &&
                             nd.last_use[out + 1] == i + 2;
            RowLnArgs a{};
            a.A = in; a.B = (o.ext ? Pext : P) + o.w; a.C = c.Y[out + (two ? 2 : 1)];
            a.M = M; a.N = o.out_dim; a.K = o.in_dim; a.lda = o.in_dim; a.ldb = o.in_dim; a.ldc = o.out_dim;
            a.bias = (o.ext ? Pext : P) + o.b; a.act = o.act;
            a.Radd = o.res >= 0 ? (o.res == 0 ? X : c.Y[o.res]) : nullptr;
            a.gamma = P + ln.w; a.beta = P + ln.b;
            if (two) { a.gamma2 = P + nd.ops[i + 2].w; a.beta2 = P + nd.ops[i + 2].b; }
            // ... and the output head behind the norm (the policy's mean / log_std layer): the 512-wide normalised
            // activation is never written
            const int lo = out + (two ? 2 : 1);  // buffer index of the last norm's output
            const int hi = i + (two ? 3 : 2);    // op index that would be the head
            bool head = false;
            if (hi < (int)nd.ops.size() && nd.ops[hi].type == OP_HEAD && nd.ops[hi].src == lo && nd.last_use[lo] == hi &&
                nd.ops[hi].out_dim <= 4) {
                const Op& ho = nd.ops[hi];
                a.headW = P + ho.w; a.headB = P + ho.b; a.headOut = c.Y[hi + 1]; a.headN = ho.out_dim;
                head = true;
            }
            launch_rowln(a, st);
            i += (two ? 2 : 1) + (head ? 1 : 0);
            continue;
        }
        // acting pass of the MLP family: hidden Linear (+ReLU) and the output head behind it in one launch (the hidden
        // activation is never written)
        if (o.type == OP_LINEAR && !save && !dc && G == 1 && M >= fuse_ln_min_rows() && !o.rowtab && o.res < 0 && (o.in_dim % GBK) == 0 &&
            (o.out_dim == 256 || o.out_dim == 512) && i + 1 < (int)nd.ops.size() && nd.ops[i + 1].type == OP_HEAD &&
            nd.ops[i + 1].src == out && nd.last_use[out] == i + 1 && nd.ops[i + 1].out_dim <= 4 && g_force_variant == 0) {
            const Op& ho = nd.ops[i + 1];
            RowLnArgs a{};
            a.A = in; a.B = (o.ext ? Pext : P) + o.w; a.C = nullptr;
            a.M = M; a.N = o.out_dim; a.K = o.in_dim; a.lda = o.in_dim; a.ldb = o.in_dim; a.ldc = o.out_dim;
            a.bias = (o.ext ? Pext : P) + o.b; a.act = o.act;
            a.headW = P + ho.w; a.headB = P + ho.b; a.headOut = c.Y[out + 1]; a.headN = ho.out_dim;
            launch_rowln(a, st);
            i += 1;
            continue;
        }
        if (o.type == OP_LINEAR) {
            GemmArgs g{};
            g.A = in; g.B = (o.ext ? Pext : P) + o.w; g.C = c.Y[out];
            g.M = M; g.N = o.out_dim; g.K = o.in_dim; g.K1 = o.in_dim;
            g.lda = o.in_dim; g.ldb = o.in_dim; g.ldc = o.out_dim;
            g.bias = (o.ext ? Pext : P) + o.b;
            if (o.rowtab && pe) { g.rowtab = pe; g.rowtab_rows = pe_rows; }
            g.Zout = (save && o.act != ACT_NONE) ? c.Z[out] : nullptr;
            g.act = o.act;
            g.Radd = o.res >= 0 ? (o.res == 0 ? X : c.Y[o.res]) : nullptr;
            g.gA = gin; g.gB = gP; g.gC = c.gY[out]; g.gBias = gP; g.gZ = c.gY[out];
            g.gR = o.res >= 0 ? (o.res == 0 ? gX : c.gY[o.res]) : 0;
            g.drop = drop_args(dc, i, o.drop);
            if (fold_n > 0) {  // A = LayerNorm(s) of the rows of the first folded norm's input
                const Op& l0 = nd.ops[fold_i];
                const int last = fold_i + fold_n - 1;
                g.A = c.Y[l0.src]; g.gA = c.gY[l0.src];
                LnA& ln = g.lnA;
                ln.gamma = P + l0.w; ln.beta = P + l0.b; ln.gP = gP; ln.gS = M;
                ln.mean = save ? c.mean[fold_i + 1] : nullptr; ln.rstd = save ? c.rstd[fold_i + 1] : nullptr;
                if (fold_n == 2) {
                    ln.gamma2 = P + nd.ops[last].w; ln.beta2 = P + nd.ops[last].b;
                    ln.mean2 = save ? c.mean[last + 1] : nullptr; ln.rstd2 = save ? c.rstd[last + 1] : nullptr;
                    ln.Y1 = c.Y[fold_i + 1];
                }
                ln.Y = c.Y[last + 1]; ln.gY = c.gY[last + 1];
                g.lnDrop = drop_args(dc, last, nd.ops[last].drop);
                launch_gemm_ln(g, G, st);
                fold_n = 0;
            } else {
                // LayerNorm(s) directly behind this Linear (and the output head behind them) finish inside its launch
                LnTail t{};
                bool head = false;
                const int nln = (skinny_fast_ok(g) && (M / 32) * G <= kLnTailCounters)
                                    ? make_ln_tail(nd, i, P, gP, M, c, save, dc, t, head) : 0;
                if (nln > 0) {
                    launch_gemm_lnt(g, t, G, st);
                    i += nln + (head ? 1 : 0);
                } else {
                    launch_gemm(true, true, g, G, st);
                }
            }
        } else if (o.type == OP_LN) {
            LnArgs a{};
            a.X = in; a.Y = c.Y[out]; a.gamma = P + o.w; a.beta = P + o.b;
            a.mean = save ? c.mean[out] : nullptr; a.rstd = save ? c.rstd[out] : nullptr;
            a.M = M; a.N = o.out_dim; a.gX = gin; a.gY = c.gY[out]; a.gP = gP; a.gS = M;
            a.drop = drop_args(dc, i, o.drop);
            bool head = false;  // the output head behind this norm rides along (its input, this norm's output, is still written)
            if ((o.out_dim == 256 || o.out_dim == 512) && i + 1 < (int)nd.ops.size() && nd.ops[i + 1].type == OP_HEAD &&
                nd.ops[i + 1].src == out && nd.ops[i + 1].out_dim <= 4) {
                const Op& ho = nd.ops[i + 1];
                a.headW = P + ho.w; a.headB = P + ho.b; a.headOut = c.Y[out + 1]; a.headN = ho.out_dim;
                a.gHW = gP; a.gHO = c.gY[out + 1];
                head = true;
            }
            dim3 grid((M + 3) / 4, G), block(256);
            if (o.out_dim == 256) hipLaunchKernelGGL((layernorm_fwd_kernel<4>), grid, block, 0, st, a);
            else if (o.out_dim == 512) hipLaunchKernelGGL((layernorm_fwd_kernel<8>), grid, block, 0, st, a);
            else hipLaunchKernelGGL(layernorm_fwd_any_kernel, grid, block, 0, st, a);
            if (head) i += 1;
        } else {
            HeadArgs a{};
            a.X = in; a.W = P + o.w; a.b = P + o.b; a.out = c.Y[out];
            a.M = M; a.K = o.in_dim; a.NO = o.out_dim; a.gX = gin; a.gW = gP; a.gB = gP; a.gO = c.gY[out];
            hipLaunchKernelGGL(head_fwd_kernel, dim3((M + 3) / 4, G), dim3(256), 0, st, a);
        }
    }
}

__global__ void head_bwd_dx_act_kernel(HeadBwdArgs a, const float* Zp, long gZ, int dact, float* colsum, long gCol) {
    // as head_bwd_dx_kernel, then dX *= act'(Zp) and column sums (the producer of X is an activated Linear)
    const long z = blockIdx.y;
    const int row = blockIdx.x;
    const float* dO = a.dOut + z * a.gD + (long)row * a.NO;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < a.NO; ++j) d[j] = dO[j];
    for (int k = threadIdx.x; k < a.K; k += blockDim.x) {
        float v = 0.f;
        for (int j = 0; j < a.NO; ++j) v += d[j] * a.W[z * a.gW + (long)j * a.K + k];
        v *= act_grad(Zp[z * gZ + (long)row * a.K + k], dact);
        a.dX[z * a.gX + (long)row * a.K + k] = v;
        if (colsum) atomicAdd(colsum + z * gCol + k, v);
    }
}

// backward of one net.  c.dY[last] must hold the gradient w.r.t. the head output.  Gr == nullptr: data
// gradients only (the critics inside the actor step).  Returns the input gradient in c.dY[0] when wanted.
__global__ void critic_dgrad_headgrad_kernel(const float* __restrict__ dZ, long gZ, const float* __restrict__ W, long gW, int N, int K,
                                             int G, int obs_dim, const float* __restrict__ head, const float* __restrict__ eps,
                                             float* __restrict__ dhead, int M, int A, float alpha);
// optional rider of the critics' data-gradient pass in the policy phase: the action columns of the input gradient are turned into
// the gradient w.r.t. the policy head's output in the same launch (critic_dgrad_headgrad_kernel)
struct HeadGradFuse { const float* head; const float* eps; float* dhead; int A, obs_dim; float alpha; bool done; };
static void net_backward(const NetDef& nd, const float* P, long gP, float* Gr, long gG, const float* X, long gX, int M, int G,
                         Ctx& c, bool want_input_grad, hipStream_t st, const In2* in2 = nullptr, const DropCtl* dc = nullptr,
                         HeadGradFuse* hg = nullptr) {
    for (int i = (int)nd.ops.size() - 1; i >= 0; --i) {
        const Op& o = nd.ops[i];
        const int out = i + 1;
        const float* in = o.src == 0 ? X : c.Y[o.src];
        const long gin = o.src == 0 ? gX : c.gY[o.src];
        const int prod = nd.producer[o.src];
        const Op* po = prod >= 0 ? &nd.ops[prod] : nullptr;
        const bool prod_lin = po && po->type == OP_LINEAR;
        const bool prod_act = prod_lin && po->act != ACT_NONE;
        float* prod_bias_grad = (Gr && prod_lin) ? Gr + po->b : nullptr;
        if (o.type == OP_HEAD) {
            HeadBwdArgs a{};
            a.dOut = c.dY[out]; a.X = in; a.W = P + o.w; a.dX = c.dY[o.src];
            a.dW = Gr ? Gr + o.w : nullptr; a.db = Gr ? Gr + o.b : nullptr;
            a.M = M; a.K = o.in_dim; a.NO = o.out_dim;
            a.gD = c.gY[out]; a.gX = gin; a.gW = gP; a.gB = gP;
            HeadBwdArgs ad = a;  // dX uses the activation-gradient stride of the source buffer
            ad.gX = c.gY[o.src];
            const bool into_ln = po && po->type == OP_LN && (po->out_dim == 256 || po->out_dim == 512);  // dX folded into the
            if (prod_act)                                                                                 // norm's backward
                hipLaunchKernelGGL(head_bwd_dx_act_kernel, dim3(M, G), dim3(256), 0, st, ad, c.Z[o.src], c.gY[o.src], po->act,
                                   prod_bias_grad, gG);
            else if (!into_ln)
                hipLaunchKernelGGL(head_bwd_dx_kernel, dim3(M, G), dim3(256), 0, st, ad);
            if (Gr) {
                HeadBwdArgs aw = a;
                aw.gW = gG; aw.gB = gG;
                hipLaunchKernelGGL(head_bwd_dw_kernel, dim3((o.out_dim * o.in_dim + 255) / 256, G, (M + 31) / 32), dim3(256), 0, st, aw);
            }
        } else if (o.type == OP_LN) {
            LnBwdArgs a{};
            a.dY = c.dY[out]; a.X = in; a.gamma = P + o.w; a.mean = c.mean[out]; a.rstd = c.rstd[out];
            a.dX = c.dY[o.src];
            a.dgamma = Gr ? Gr + o.w : nullptr; a.dbeta = Gr ? Gr + o.b : nullptr;
            a.Zp = prod_act ? c.Z[o.src] : nullptr; a.dact = prod_act ? po->act : 0;
            a.colsum = prod_bias_grad;
            a.M = M; a.N = o.out_dim;
            a.gA = c.gY[out]; a.gP = Gr ? gG : gP; a.gS = M;
            a.dmask = drop_args(dc, i, o.drop);  // this norm's output was dropped: mask the incoming gradient
            if ((o.out_dim == 256 || o.out_dim == 512) && i + 1 < (int)nd.ops.size() && nd.ops[i + 1].type == OP_HEAD &&
                nd.ops[i + 1].src == out) {  // dY = (head's output gradient) x (head weights), evaluated in place
                const Op& ho = nd.ops[i + 1];
                a.hdOut = c.dY[out + 1]; a.hW = P + ho.w; a.hN = ho.out_dim; a.gHD = c.gY[out + 1]; a.gHW = gP;
            }
            if (dc && prod_lin && po->drop && po->res >= 0) {  // producer = Linear -> dropout -> + residual: its dZ is the masked dX
                a.dXm = c.dYm[o.src];
                a.omask = drop_args(dc, prod, po->drop);
            }
            // gamma is read with the PARAMETER stride, dgamma written with the GRADIENT stride: both nets use the
            // same block layout, so the strides coincide whenever Gr != nullptr (gG == gP is asserted at create)
            dim3 grid((M + 4 * LN_BWD_RPW - 1) / (4 * LN_BWD_RPW), G), block(256);
            if (o.out_dim == 256) hipLaunchKernelGGL((layernorm_bwd_kernel<4>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((layernorm_bwd_kernel<8>), grid, block, 0, st, a);
        } else {  // LINEAR: c.dY[out] already holds dZ (act' and bias column sums were fused by its writer)
            const float* dZ = (dc && o.drop && o.res >= 0) ? c.dYm[out] : c.dY[out];
            if (thin_ok(o) && o.src == 0 && g_force_variant == 0) {
                ThinArgs a = thin_input(o, X, gX, in2, M);
                a.dZ = dZ; a.gY = c.gY[out];
                if (Gr) {
                    a.dW = Gr + o.w; a.gDW = gG;
                    // (the gradient buffer is zero here: tvc_sac_*_grads memset it or the Adam kernel left it zeroed)
                    hipLaunchKernelGGL(thin_wgrad_kernel, dim3((o.out_dim + 31) / 32, thin_wgrad_slices(M), G), dim3(256), 0, st, a);
                }
                if (want_input_grad && hg && hg->A <= 2 && o.in_dim == hg->obs_dim + hg->A) {
                    hipLaunchKernelGGL(critic_dgrad_headgrad_kernel, dim3((M + 3) / 4), dim3(256), 0, st, dZ, c.gY[out], P + o.w, gP,
                                       o.out_dim, o.in_dim, G, hg->obs_dim, hg->head, hg->eps, hg->dhead, M, hg->A, hg->alpha);
                    hg->done = true;
                } else if (want_input_grad) {
                    a.W = P + o.w; a.gW = gP; a.dX = c.dY[0]; a.gDX = c.gY[0];
                    hipLaunchKernelGGL(thin_dgrad_kernel, dim3((M + 3) / 4, 1, G), dim3(256), 0, st, a);
                }
                continue;
            }
            GemmArgs gw{};
            if (Gr) {  // dW[out,in] = dZ^T X
                GemmArgs& g = gw;
                g.A = dZ; g.B = in; g.C = Gr + o.w;
                g.M = o.out_dim; g.N = o.in_dim; g.K = M; g.K1 = M;
                g.lda = o.out_dim; g.ldb = o.in_dim; g.ldc = o.in_dim;
                g.gA = c.gY[out]; g.gB = gin; g.gC = gG;
                if (!(o.src > 0 || want_input_grad)) launch_gemm(false, false, g, G, st);
            }
            if (o.src > 0 || want_input_grad) {  // dX[M,in] = dZ W (+ residual-branch gradient) (* act' of the producer)
                GemmArgs g{};
                g.A = dZ; g.B = P + o.w; g.C = c.dY[o.src];
                g.M = M; g.N = o.in_dim; g.K = o.out_dim; g.K1 = o.out_dim;
                g.lda = o.out_dim; g.ldb = o.in_dim; g.ldc = o.in_dim;
                g.gA = c.gY[out]; g.gB = gP; g.gC = c.gY[o.src];
                const int rc = nd.res_consumer[o.src];
                if (rc >= 0 && rc != i) { g.Radd = c.dY[rc + 1]; g.gR = c.gY[rc + 1]; }
                if (prod_act) { g.dactZ = c.Z[o.src]; g.dact = po->act; g.gDZ = c.gY[o.src]; }
                if (prod_bias_grad) { g.colsum = prod_bias_grad; g.gCol = gG; }
                if (dc && prod_lin && po->drop && po->res < 0) g.dmask = drop_args(dc, prod, po->drop);  // producer's own dropout
                if (Gr) launch_gemm_bwd_pair(gw, g, G, st);
                else launch_gemm(true, false, g, G, st);
            }
        }
    }
}

// ------------------------------------------------------------------ SAC elementwise kernels
__global__ void concat_kernel(const float* __restrict__ s, const float* __restrict__ a, float* __restrict__ x, int M, int no,
                              int na) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * (no + na)) return;
    const int m = i / (no + na), k = i - m * (no + na);
    x[i] = k < no ? s[(long)m * no + k] : a[(long)m * na + (k - no)];
}

// head output [M,2A] -> mean, clamped log_std, action = mean + exp(log_std) * eps  (agent/...:224-225, 780-782, 964-966)
__global__ void adam_tick_kernel(AdamClock* clk, float b1, float b2) { adam_clock_advance(clk, b1, b2); }  // (rare: flush paths)
__global__ void act_ctr_tick_kernel(int* ctr) { *ctr += 1; }
__global__ void sample_action_kernel(const float* __restrict__ head, const float* __restrict__ eps, float* __restrict__ act,
                                     float* __restrict__ mean_out, float* __restrict__ ls_out, int M, int A, int clamp_act,
                                     Ticks tk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        if (tk.clk) adam_clock_advance(tk.clk, tk.b1, tk.b2);
        if (tk.ctr) *tk.ctr += 1;
    }
    if (i >= M * A) return;
    const int m = i / A, j = i - m * A;
    const float mu = head[(long)m * 2 * A + j];
    const float ls = fminf(fmaxf(head[(long)m * 2 * A + A + j], -20.0f), 2.0f);
    float a = eps ? mu + expf(ls) * eps[i] : mu;
    if (clamp_act) a = fminf(fmaxf(a, -1.0f), 1.0f);
    act[i] = a;
    if (mean_out) mean_out[i] = mu;
    if (ls_out) ls_out[i] = ls;
}


__device__ __forceinline__ float block_sum(float v) {
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}

// y = r + gamma (1 - d) min(tq1, tq2) (agent/...:968-971); q [2,M] -> dq = 2 (q - y) / M, losses[g] += mean((q - y)^2) (:976-977)
__global__ void __launch_bounds__(256) q_loss_kernel(const float* __restrict__ q, const float* __restrict__ tq,
                                                     const float* __restrict__ r, const float* __restrict__ d, float gamma,
                                                     float* __restrict__ dq, float* __restrict__ losses, int M) {
    const int g = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    float e = 0.f;
    if (m < M) {
        const float y = r[m] + gamma * (1.0f - d[m]) * fminf(tq[m], tq[M + m]);
        const float diff = q[(long)g * M + m] - y;
        dq[(long)g * M + m] = 2.0f * diff / (float)M;
        e = diff * diff / (float)M;
    }
    const float s = block_sum(e);
    if (threadIdx.x == 0) atomicAdd(&losses[g], s);
}

// policy loss and its gradient w.r.t. the two critic outputs (agent/...:994-998)
__global__ void __launch_bounds__(256) actor_loss_kernel(const float* __restrict__ qn, const float* __restrict__ ls,
                                                         const float* __restrict__ eps, float* __restrict__ dqn,
                                                         float* __restrict__ losses, int M, int A, float alpha) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    float e = 0.f;
    if (m < M) {
        float logp = 0.f;
        for (int j = 0; j < A; ++j) {
            const float ep = eps[(long)m * A + j];
            logp += -0.5f * ep * ep - ls[(long)m * A + j] - 0.9189385332046727f;  // log sqrt(2 pi)
        }
        const float q1 = qn[m], q2 = qn[M + m];
        const bool first = q1 <= q2;
        e = -(fminf(q1, q2) - alpha * logp) / (float)M;
        dqn[m] = first ? -1.0f / (float)M : 0.0f;
        dqn[M + m] = first ? 0.0f : -1.0f / (float)M;
    }
    const float s = block_sum(e);
    if (threadIdx.x == 0) atomicAdd(&losses[2], s);
}

// gradient w.r.t. the policy head output [M,2A] from d(a_new) (two critics' input gradients) and the entropy term
__global__ void actor_head_grad_kernel(const float* __restrict__ dxin, long g_stride, int in_dim, int obs_dim,
                                       const float* __restrict__ head, const float* __restrict__ eps,
                                       float* __restrict__ dhead, int M, int A, float alpha) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * A) return;
    const int m = i / A, j = i - m * A;
    const float da = dxin[(long)m * in_dim + obs_dim + j] + dxin[g_stride + (long)m * in_dim + obs_dim + j];
    const float raw = head[(long)m * 2 * A + A + j];
    const float ls = fminf(fmaxf(raw, -20.0f), 2.0f);
    const bool pass = raw >= -20.0f && raw <= 2.0f;  // torch.clamp passes the gradient on [min, max]
    dhead[(long)m * 2 * A + j] = da;
    dhead[(long)m * 2 * A + A + j] = pass ? (da * expf(ls) * eps[i] - alpha / (float)M) : 0.0f;
}

// ... and the same fed straight from the critics' first-layer gradients: da[m, j] = sum over both critics and their hidden units
// of dZ[m, n] W[n, obs + j] (only the ACTION columns of the critics' input gradient are ever used), one wave per row; replaces
// the thin dgrad launch + actor_head_grad_kernel of the policy phase
__global__ void __launch_bounds__(256) critic_dgrad_headgrad_kernel(const float* __restrict__ dZ, long gZ, const float* __restrict__ W,
                                                                    long gW, int N, int K, int G, int obs_dim,
                                                                    const float* __restrict__ head, const float* __restrict__ eps,
                                                                    float* __restrict__ dhead, int M, int A, float alpha) {
    TVC_LEARNER_PRIO();
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float acc[2] = {0.f, 0.f};
    for (int z = 0; z < G; ++z) {
        const float* dz = dZ + z * gZ + (long)m * N;
        const float* w = W + z * gW;
        for (int n = lane; n < N; n += 64) {
            const float v = dz[n];
            for (int j = 0; j < A; ++j) acc[j] = fmaf(v, w[(long)n * K + obs_dim + j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
    if (lane < A) {
        const int j = lane;
        const float da = j == 0 ? acc[0] : acc[1];
        const float raw = head[(long)m * 2 * A + A + j];
        const float ls = fminf(fmaxf(raw, -20.0f), 2.0f);
        const bool pass = raw >= -20.0f && raw <= 2.0f;
        dhead[(long)m * 2 * A + j] = da;
        dhead[(long)m * 2 * A + A + j] = pass ? (da * expf(ls) * eps[(long)m * A + j] - alpha / (float)M) : 0.0f;
    }
}

// First kernel of an update.  Workgroup 0: losses[0..2] = 0 and losses[3] = PhysicsInformedLoss.forward
// (agent/...:236-285, reported only); the other workgroups stack xs2 = [s ; s'] for the single actor forward.
__global__ void __launch_bounds__(256) update_prep_kernel(const float* __restrict__ s, const float* __restrict__ a,
                                                          const float* __restrict__ s2, float* __restrict__ losses,
                                                          float* __restrict__ xs2, int M, int no, int na, float weight,
                                                          float* __restrict__ acat) {
    if (blockIdx.x > 0) {
        const long n = (long)M * no;
        for (long i = (long)(blockIdx.x - 1) * 256 + threadIdx.x; i < 2 * n; i += (long)(gridDim.x - 1) * 256)
            xs2[i] = i < n ? s[i] : s2[i - n];
        // the batch's actions in front of the slot the target actions a' ~ pi(s') are sampled into: [a ; a'] is the second input
        // source of the ONE critic forward over (s, a) and (s', a') (tvc_sac_critic_grads)
        if (acat)
            for (long i = (long)(blockIdx.x - 1) * 256 + threadIdx.x; i < (long)M * na; i += (long)(gridDim.x - 1) * 256) acat[i] = a[i];
        return;
    }
    float e = 0.f;
    for (int m = threadIdx.x; m < M && no >= 7; m += 256) {
        const float* x = s + (long)m * no;
        const float* x2 = s2 + (long)m * no;
        float an2 = 0.f;
        for (int j = 0; j < na; ++j) an2 += a[(long)m * na + j] * a[(long)m * na + j];
        const float ctrl = sqrtf(an2) * 0.1f;
        float mom = 0.f, ke = 0.f, ke2 = 0.f;
        for (int j = 4; j < 7; ++j) {
            const float d = x2[j] - (x[j] + ctrl);
            mom += d * d;
            ke += x[j] * x[j];
            ke2 += x2[j] * x2[j];
        }
        const float de = 0.5f * ke2 - (0.5f * ke + 0.5f * an2 * 0.01f);
        const float qn = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]) - 1.0f;
        const float qn2 = sqrtf(x2[0] * x2[0] + x2[1] * x2[1] + x2[2] * x2[2] + x2[3] * x2[3]) - 1.0f;
        e += weight * (mom / (3.0f * M) + de * de / (float)M + (qn * qn + qn2 * qn2) / (float)M);
    }
    const float sum = block_sum(e);
    if (threadIdx.x == 0) { losses[0] = 0.f; losses[1] = 0.f; losses[2] = 0.f; losses[3] = sum; }
}

// torch.optim.Adam defaults written out (agent/...:623-625): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps); g is pre-scaled by gscale (1 / world size).
// t and the powers come from the device-resident clock (AdamClock above: this launch is step clk->step + 1).
// The gradient is zeroed once consumed, so the next update needs no memset launch.
// Workgroups from n_adam_blocks on run the Polyak update of the target critics (agent/...:1005-1010) beside the optimiser step:
// it reads the online critics (stepped by an EARLIER launch) and touches nothing the Adam role does.
__global__ void __launch_bounds__(256) adam_dev_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                       const AdamClock* __restrict__ clk, float gscale, int n_adam_blocks,
                                                       float* __restrict__ pol_tgt, const float* __restrict__ pol_src, long pol_n,
                                                       float tau) {
    TVC_LEARNER_PRIO();
    if ((int)blockIdx.x >= n_adam_blocks) {
        const long stride = (long)(gridDim.x - n_adam_blocks) * blockDim.x;
        for (long i = (long)(blockIdx.x - n_adam_blocks) * blockDim.x + threadIdx.x; i < pol_n; i += stride)
            pol_tgt[i] = tau * pol_src[i] + (1.0f - tau) * pol_tgt[i];
        return;
    }
    __shared__ float bc[2];
    if (threadIdx.x == 0) {
        const double b1t = clk->b1t * (double)b1, b2t = clk->b2t * (double)b2;
        bc[0] = (float)(1.0 - b1t);
        bc[1] = (float)sqrt(1.0 - b2t);
    }
    __syncthreads();
    const float bc1 = bc[0], bc2s = bc[1];
    const long n4 = n >> 2;  // all four arrays are 16-byte aligned (asserted at create); n is a multiple of 4
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)n_adam_blocks * blockDim.x) {
        float4 g4 = reinterpret_cast<float4*>(g)[i], m4 = reinterpret_cast<float4*>(m)[i];
        float4 v4 = reinterpret_cast<float4*>(v)[i], p4 = reinterpret_cast<float4*>(p)[i];
        float* gp = &g4.x; float* mp = &m4.x; float* vp = &v4.x; float* pp = &p4.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gi = gp[j] * gscale;
            const float mi = b1 * mp[j] + (1.0f - b1) * gi;
            const float vi = b2 * vp[j] + (1.0f - b2) * gi * gi;
            mp[j] = mi; vp[j] = vi;
            pp[j] -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
        }
        reinterpret_cast<float4*>(m)[i] = m4;
        reinterpret_cast<float4*>(v)[i] = v4;
        reinterpret_cast<float4*>(p)[i] = p4;
        reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

}  // namespace

// ------------------------------------------------------------------ handle
struct tvc_sac {
    tvc_sac_cfg cfg;
    int device;
    NetDef actor, critic, actor_inf;
    FoldInfo fold;
    float* ov = nullptr;  // [layers][d*d + d] folded attention weights of the acting net
    float *snap_p = nullptr, *snap_ov = nullptr;  // acting snapshot of the policy parameters / folded weights
    long ov_floats = 0;
    long n_actor, n_critic;
    float *params, *grads, *adam_m, *adam_v;  // caller-owned
    void* slab = nullptr;                      // library-owned workspace
    Ctx actx, cctx, ictx;                      // actor (train), critics (train, G=2), actor (inference, slot-aliased)
    Ctx dctx;                                  // actor as trained (unfolded, dropout sites), slot-aliased: acting in train mode
    bool dctx_ok = false;
    long head_off[4] = {0, 0, 0, 0};           // policy_head.6 weight / bias, policy_head.8 weight / bias (pack_head_kernel)
    int* act_ctr = nullptr;                    // dropout counter of the train-mode acting calls (advanced by each call)
    float *xs2 = nullptr;   // [2B, obs]: states and next states stacked for the single actor forward of an update
    bool actor_fwd_valid = false;
    bool grads_clean[2] = {false, false};  // critics, actor: zeroed by the Adam kernel since they were last written
    float *acat = nullptr;  // [2, B, A]: the batch's actions, then a_tmp (sampled actions)
    float *pe = nullptr, *xcat = nullptr, *a_tmp = nullptr, *y = nullptr, *dq = nullptr, *ls_tmp = nullptr, *mean_tmp = nullptr;
    AdamClock* clk = nullptr;  // [2]: critics, actor
    // one-launch acting pass (tvc_actor_rows.h): packed weight-tile stream + vector section, rebuilt after every policy update
    bool rows_ok = false;
    int rows_tiles = 0, rows_vecs = 0;
    long pack_floats = 0;
    float *pack = nullptr, *snap_pack = nullptr;  // [rows_tiles * 4096 tile floats | vector section]
    PackTile* d_ptiles = nullptr;
    PackVec* d_pvecs = nullptr;
    // ... and the stream of the net AS TRAINED (attention and embedding not folded) for train-mode acting (actor_split_kernel<true>)
    bool train_rows_ok = false;
    bool train_stream_live = false;  // packed (and re-packed by every policy update, copied by every snapshot) only once train-mode
                                     // acting has been asked for: the default loop does not pay for a stream it never reads
    int trows_tiles = 0, trows_vecs = 0;
    long tpack_floats = 0;
    float *tpack = nullptr, *snap_tpack = nullptr;  // [trows_tiles * 4096 tile floats | train vector section]
    PackTile* d_tptiles = nullptr;
    PackVec* d_tpvecs = nullptr;
    // ... and the folded net as bf16 triples for the split-operand acting kernel (tvc_actor_x3.h, tvc_sac_act flags bit 4): packed,
    // re-packed and snapshotted only once that kernel has been asked for
    bool x3_live = false, x3_attr_set = false;
    int x3_tiles = 0;
    char *xpack = nullptr, *snap_xpack = nullptr;  // [x3_tiles * 24 KB]; the vector section is `pack`'s
    PackTile3* d_xtiles = nullptr;
    bool x3t_live = false, x3t_attr_set = false;   // ... and of the net as trained (train-mode acting, flags bits 3 + 4)
    int x3t_tiles = 0;
    char *xtpack = nullptr, *snap_xtpack = nullptr;
    PackTile3* d_xttiles = nullptr;
    float* tq = nullptr;                          // [2, B]: output of the target critics (read by q_loss_kernel)
    bool tick_pending = false;                    // the critics' Adam clock is one step behind: its advance rides on the next
                                                  // launch of the chain (tvc_sac_actor_grads), or is flushed by whoever needs it
    bool lds_attr_set = false;                    // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for this handle's device
    bool split_attr_set = false, tsplit_attr_set = false;  // ... for actor_split_kernel<false> / <true>
    unsigned long long* rows_stamps = nullptr;    // diagnostics: set by rows_probe around its launches
    float* P_actor() { return params; }
    float* P_q() { return params + n_actor; }
    float* P_tq() { return params + n_actor + 2 * n_critic; }
    float* G_actor() { return grads; }
    float* G_q() { return grads + n_actor; }
};

// The acting megakernel covers the reference shapes with the embedding folded (PE(0) on every row): everything else keeps
// the per-layer kernels.
static bool rows_supported(const tvc_sac_cfg& c, const FoldInfo& f) {
    return c.family == 0 && f.embed && c.d_model == 256 && c.ff_dim == 512 && c.head1 == 512 && c.head2 == 512 &&
           2 * c.act_dim <= 4 && c.obs_dim <= 16 && f.layers == c.n_layers && c.pe_rows == 1;
}
static int rows_min_rows() {  // acting batches at least this large take the one-launch path with 64 rows per workgroup
    static const int v = [] { const char* e = getenv("TVC_ROWS_MIN"); return e ? atoi(e) : 16384; }();
    return v;
}
static int split_min_rows() {  // ... and from here up to rows_min_rows() the one-launch path with 16 rows per workgroup (tvc_actor_split.h)
    static const int v = [] { const char* e = getenv("TVC_SPLIT_MIN"); return e ? atoi(e) : 1024; }();
    return v;
}
// Descriptor tables of pack_actor_kernel: the tile stream in the order actor_rows_kernel consumes it, and the vector section.
static void rows_tables(const tvc_sac_cfg& c, const NetDef& actor, const FoldInfo& f, std::vector<PackTile>& tiles,
                        std::vector<PackVec>& vecs) {
    auto off = [&](const std::string& name) -> long {
        for (const TensorInfo& t : actor.tensors)
            if (t.name == name) return t.off;
        return -1;
    };
    const int d = 256;
    const long ostride = (long)d * d + d;
    auto pass = [&](long src, int ld, int n0, int k0, int ktiles, int kvalid, int from_ov) {
        for (int kt = 0; kt < ktiles; ++kt) tiles.push_back(PackTile{src + (long)n0 * ld, ld, k0 + 16 * kt, kvalid, from_ov, 0});
    };
    // deep tiles (32 k x 128 output rows n0 .. n0 + 127), kt2 of them = 32 kt2 inputs
    auto deep = [&](long src, int ld, int n0, int kt2, int kvalid) {
        for (int kt = 0; kt < kt2; ++kt) tiles.push_back(PackTile{src + (long)n0 * ld, ld, 32 * kt, kvalid, 0, 2});
    };
    auto vec = [&](long src, int dst, int count, int from_ov) { vecs.push_back(PackVec{src, dst, count, from_ov}); };
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string p = "layers." + std::to_string(l) + ".";
        const int lv = l * AR_LAYER_VEC;
        if (l == 0) {
            pass(f.e_off, f.obs, 0, 0, 2, f.obs, 1);                         // W' [256][obs]: one zero-padded 16-deep tile + an all-zero one
            vec(f.e_off + (long)d * f.obs, lv, d, 1);                          // b'
        } else {
            pass(l * ostride, d, 0, 0, 16, d, 1);                              // W_ov of layer l
            vec(l * ostride + (long)d * d, lv, d, 1);                          // b_ov
        }
        vec(off(p + "norm1.weight"), lv + 256, d, 0);
        vec(off(p + "norm1.bias"), lv + 512, d, 0);
        for (int quarter = 0; quarter < 4; ++quarter) {
            deep(off(p + "linear1.weight"), d, 128 * quarter, 8, d);              // hidden units [128 quarter, +128) <- x: 8 deep tiles
            pass(off(p + "linear2.weight"), 512, 0, 128 * quarter, 8, 512, 0);    // out += W2[:, hidden quarter] h
        }
        vec(off(p + "linear1.bias"), lv + 768, 512, 0);
        vec(off(p + "linear2.bias"), lv + 1280, d, 0);
        vec(off(p + "norm2.weight"), lv + 1536, d, 0);
        vec(off(p + "norm2.bias"), lv + 1792, d, 0);
    }
    const int tv = c.n_layers * AR_LAYER_VEC;
    vec(off("feature_norm.weight"), tv, d, 0);
    vec(off("feature_norm.bias"), tv + 256, d, 0);
    if (c.use_se) {  // SqueezeExcitation: fc1 [16][256] as one blocked tile + a zero tile (k0 beyond kvalid), fc2 [256][16] like the embedding
        tiles.push_back(PackTile{off("se_block.fc1.weight"), d, 0, d, 0, 1});
        tiles.push_back(PackTile{off("se_block.fc1.weight"), d, d, d, 0, 0});
        pass(off("se_block.fc2.weight"), 16, 0, 0, 2, 16, 0);
        vec(off("se_block.fc1.bias"), tv + AR_TAIL_VEC, 16, 0);
        vec(off("se_block.fc2.bias"), tv + AR_TAIL_VEC + 16, d, 0);
    }
    for (int half = 0; half < 2; ++half) pass(off("policy_head.0.weight"), d, 256 * half, 0, 16, d, 0);
    vec(off("policy_head.0.bias"), tv + 512, 512, 0);
    vec(off("policy_head.2.weight"), tv + 1024, 512, 0);
    vec(off("policy_head.2.bias"), tv + 1536, 512, 0);
    for (int half = 0; half < 2; ++half) pass(off("policy_head.4.weight"), 512, 256 * half, 0, 32, 512, 0);
    vec(off("policy_head.4.bias"), tv + 2048, 512, 0);
    // policy_head.6 (LayerNorm) and policy_head.8 (output Linear) enter the tail through pack_head_kernel, already folded
}

// The same net as the split-operand stream (tvc_actor_x3.h): per pass NT n-tiles x KB k-blocks, 8 triples per tile.
static void rows_tables_x3(const tvc_sac_cfg& c, const NetDef& actor, const FoldInfo& f, std::vector<PackTile3>& tiles) {
    auto off = [&](const std::string& name) -> long {
        for (const TensorInfo& t : actor.tensors)
            if (t.name == name) return t.off;
        return -1;
    };
    const int d = 256;
    const long ostride = (long)d * d + d;
    auto pass = [&](long src, int ld, int n0, int k0, int NT, int KB, int kvalid, int nvalid, int from_ov) {
        for (int g0 = 0; g0 < NT * KB; g0 += X3_TRI)
            tiles.push_back(PackTile3{src < 0 ? -1 : src + (long)n0 * ld, ld, k0, kvalid, nvalid, g0, NT, from_ov});
    };
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string p = "layers." + std::to_string(l) + ".";
        if (l == 0) pass(f.e_off, f.obs, 0, 0, 16, 1, f.obs, d, 1);   // W' [256][obs]: one k-block, 16 triples = two tiles
        else pass(l * ostride, d, 0, 0, 16, 8, d, d, 1);               // W_ov of layer l
        for (int quarter = 0; quarter < 4; ++quarter) {
            pass(off(p + "linear1.weight"), d, 128 * quarter, 0, 8, 8, d, 128, 0);        // hidden units [128 quarter, +128) <- x
            pass(off(p + "linear2.weight"), 512, 0, 128 * quarter, 16, 4, 512, d, 0);    // out += W2[:, hidden quarter] h
        }
    }
    if (c.use_se) {
        pass(off("se_block.fc1.weight"), d, 0, 0, 1, 8, d, 16, 0);   // [16][256]: one tile
        tiles.push_back(PackTile3{0, 1, 0, 0, 0, 0, 1, 0});          // + an all-zero tile (ring parity)
        pass(off("se_block.fc2.weight"), 16, 0, 0, 16, 1, 16, d, 0); // [256][16]: two tiles
    }
    for (int half = 0; half < 2; ++half) pass(off("policy_head.0.weight"), d, 256 * half, 0, 16, 8, d, d, 0);
    for (int part = 0; part < X3_HQ; ++part)
        pass(off("policy_head.4.weight"), 512, (512 / X3_HQ) * part, 0, 32 / X3_HQ, 16, 512, 512 / X3_HQ, 0);
}

static void rows_tables_x3_train(const tvc_sac_cfg& c, const NetDef& actor, std::vector<PackTile3>& tiles) {
    auto off = [&](const std::string& name) -> long {
        for (const TensorInfo& t : actor.tensors)
            if (t.name == name) return t.off;
        return -1;
    };
    const int d = 256;
    auto pass = [&](long src, int ld, int n0, int k0, int NT, int KB, int kvalid, int nvalid) {
        for (int g0 = 0; g0 < NT * KB; g0 += X3_TRI)
            tiles.push_back(PackTile3{src < 0 ? -1 : src + (long)n0 * ld, ld, k0, kvalid, nvalid, g0, NT, 0});
    };
    pass(off("input_embedding.weight"), c.obs_dim, 0, 0, 16, 1, c.obs_dim, d);
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string p = "layers." + std::to_string(l) + ".";
        pass(off(p + "v_proj.weight"), d, 0, 0, 16, 8, d, d);
        pass(off(p + "out_proj.weight"), d, 0, 0, 16, 8, d, d);
        for (int quarter = 0; quarter < 4; ++quarter) {
            pass(off(p + "linear1.weight"), d, 128 * quarter, 0, 8, 8, d, 128);
            pass(off(p + "linear2.weight"), 512, 0, 128 * quarter, 16, 4, 512, d);
        }
    }
    for (int half = 0; half < 2; ++half) pass(off("policy_head.0.weight"), d, 256 * half, 0, 16, 8, d, d);
    for (int part = 0; part < X3_HQ; ++part)
        pass(off("policy_head.4.weight"), 512, (512 / X3_HQ) * part, 0, 32 / X3_HQ, 16, 512, 512 / X3_HQ);
}

// Train-mode stream: embedding (one tile), then per encoder layer v_proj (16 tiles), out_proj (16), the FFN as in rows_tables(),
// then the head.  Vector section: +0 b_e, per layer l at 256 + 512 l: b_v, b_o; behind the layers beta6 x W8 [4][512] and b8[4]
// (pack_head with tail_t).
static long train_vec_floats(const tvc_sac_cfg& c) { return 256 + 512L * c.n_layers + 4 * 512 + 16; }
static void rows_tables_train(const tvc_sac_cfg& c, const NetDef& actor, std::vector<PackTile>& tiles, std::vector<PackVec>& vecs) {
    auto off = [&](const std::string& name) -> long {
        for (const TensorInfo& t : actor.tensors)
            if (t.name == name) return t.off;
        return -1;
    };
    const int d = 256;
    auto pass = [&](long src, int ld, int n0, int k0, int ktiles, int kvalid) {
        for (int kt = 0; kt < ktiles; ++kt) tiles.push_back(PackTile{src + (long)n0 * ld, ld, k0 + 16 * kt, kvalid, 0, 0});
    };
    auto deep = [&](long src, int ld, int n0, int kt2, int kvalid) {
        for (int kt = 0; kt < kt2; ++kt) tiles.push_back(PackTile{src + (long)n0 * ld, ld, 32 * kt, kvalid, 0, 2});
    };
    pass(off("input_embedding.weight"), c.obs_dim, 0, 0, 1, c.obs_dim);
    vecs.push_back(PackVec{off("input_embedding.bias"), 0, d, 0});
    for (int l = 0; l < c.n_layers; ++l) {
        const std::string p = "layers." + std::to_string(l) + ".";
        pass(off(p + "v_proj.weight"), d, 0, 0, 16, d);
        pass(off(p + "out_proj.weight"), d, 0, 0, 16, d);
        for (int quarter = 0; quarter < 4; ++quarter) {
            deep(off(p + "linear1.weight"), d, 128 * quarter, 8, d);
            pass(off(p + "linear2.weight"), 512, 0, 128 * quarter, 8, 512);
        }
        vecs.push_back(PackVec{off(p + "v_proj.bias"), 256 + 512 * l, d, 0});
        vecs.push_back(PackVec{off(p + "out_proj.bias"), 512 + 512 * l, d, 0});
    }
    for (int half = 0; half < 2; ++half) pass(off("policy_head.0.weight"), d, 256 * half, 0, 16, d);
    for (int half = 0; half < 2; ++half) pass(off("policy_head.4.weight"), 512, 256 * half, 0, 32, 512);
}

static long ctx_bytes(const NetDef& nd, int M, int G, bool train) {
    long f = 0;
    for (size_t b = 1; b < nd.buf_dim.size(); ++b) {
        f += (long)G * M * nd.buf_dim[b] * (train ? 4 : 0);  // Y, dY, Z (+ dYm where a dropped Linear feeds a residual)
        f += train ? 2L * G * M : 0;                         // mean, rstd
    }
    f += (long)G * M * nd.buf_dim[0];  // dY[0]
    f += train ? kLnTailCounters : 0;
    return f * 4;
}
static char* carve(char*& p, long bytes) {
    char* r = p;
    p += (bytes + 255) & ~255L;
    return r;
}
static void ctx_alloc_train(Ctx& c, const NetDef& nd, int M, int G, char*& p) {
    const size_t nb = nd.buf_dim.size();
    c.M = M; c.G = G;
    c.Y.assign(nb, nullptr); c.dY.assign(nb, nullptr); c.Z.assign(nb, nullptr); c.dYm.assign(nb, nullptr);
    c.mean.assign(nb, nullptr); c.rstd.assign(nb, nullptr); c.gY.assign(nb, 0);
    c.cnt = (unsigned*)carve(p, kLnTailCounters * 4);  // (the slab is zeroed at creation; every launch leaves them at zero)
    for (size_t b = 0; b < nb; ++b) {
        const long n = (long)M * nd.buf_dim[b];
        c.gY[b] = n;
        c.dY[b] = (float*)carve(p, n * G * 4);
        if (b == 0) continue;
        const Op& po = nd.ops[b - 1];
        if (po.type == OP_LINEAR && po.drop && po.res >= 0) c.dYm[b] = (float*)carve(p, n * G * 4);
        c.Y[b] = (float*)carve(p, n * G * 4);
        c.Z[b] = (float*)carve(p, n * G * 4);
        c.mean[b] = (float*)carve(p, (long)M * G * 4);
        c.rstd[b] = (float*)carve(p, (long)M * G * 4);
    }
}
// inference: activations rotate through a few slots (a buffer's slot is reused once its last consumer ran)
static int ctx_alloc_infer(Ctx& c, const NetDef& nd, int M, char*& p, int max_slots = 8) {
    const size_t nb = nd.buf_dim.size();
    c.M = M; c.G = 1;
    c.Y.assign(nb, nullptr); c.dY.assign(nb, nullptr); c.Z.assign(nb, nullptr);
    c.mean.assign(nb, nullptr); c.rstd.assign(nb, nullptr); c.gY.assign(nb, 0);
    int maxd = 0;
    for (size_t b = 1; b < nb; ++b) maxd = std::max(maxd, nd.buf_dim[b]);
    std::vector<float*> slot(max_slots);
    std::vector<int> free_at(max_slots, -1000);  // op index of the last reader of the slot (never used: far past)
    for (int s = 0; s < max_slots; ++s) slot[s] = (float*)carve(p, (long)M * maxd * 4);
    for (size_t b = 1; b < nb; ++b) {
        const int prod = (int)b - 1;
        int pick = -1;
        for (int s = 0; s < max_slots; ++s)
            // reusable once its last reader ran at least two ops earlier: a fused Linear+LN(+LN) launch writes the
            // buffer of op i+2 while it still reads the inputs of op i
            if (free_at[s] < prod - 2) { pick = s; break; }
        if (pick < 0) return -1;
        c.Y[b] = slot[pick];
        c.gY[b] = (long)M * nd.buf_dim[b];
        free_at[pick] = nd.last_use[b];
    }
    return 0;
}

extern "C" {

void tvc_sac_default_cfg(tvc_sac_cfg* c, int32_t family) {
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->obs_dim = 10; c->act_dim = 2; c->family = family;
    c->d_model = 256; c->n_layers = 4; c->ff_dim = 512; c->head1 = 512; c->head2 = 512;  // agent/...:451-468
    c->mlp1 = 256; c->mlp2 = 256;
    c->critic1 = family == 0 ? 512 : 256; c->critic2 = 256;                              // agent/...:593-603
    c->batch_size = 256; c->max_act_rows = 65536; c->pe_rows = 1;
    c->dropout_p = 0.0f; c->nhead = 8;
    c->gamma = 0.99f; c->alpha = 0.2f; c->tau = 0.005f; c->lr = 3e-4f;                   // agent/...:971,998,1005,623
    c->adam_b1 = 0.9f; c->adam_b2 = 0.999f; c->adam_eps = 1e-8f;
}

int64_t tvc_sac_param_count(const tvc_sac_cfg* c) {
    if (validate(c)) return -1;
    return build_actor(*c).n_params + 4 * build_critic(*c).n_params;
}
int64_t tvc_sac_trainable_count(const tvc_sac_cfg* c) {
    if (validate(c)) return -1;
    return build_actor(*c).n_params + 2 * build_critic(*c).n_params;
}
int32_t tvc_sac_num_tensors(const tvc_sac_cfg* c) {
    if (validate(c)) return -1;
    return (int32_t)(build_actor(*c).tensors.size() + 4 * build_critic(*c).tensors.size());
}
int tvc_sac_tensor_info(const tvc_sac_cfg* c, int32_t idx, char* name, int32_t cap, int64_t* offset, int32_t* rows, int32_t* cols) {
    if (int e = validate(c)) return e;
    NetDef a = build_actor(*c), q = build_critic(*c);
    const char* prefixes[5] = {"policy.", "q1.", "q2.", "target_q1.", "target_q2."};
    long base = 0;
    int i = idx;
    for (int net = 0; net < 5; ++net) {
        const NetDef& nd = net == 0 ? a : q;
        if (i < (int)nd.tensors.size()) {
            const TensorInfo& t = nd.tensors[i];
            if (name && cap > 0) snprintf(name, cap, "%s%s", prefixes[net], t.name.c_str());
            if (offset) *offset = base + t.off;
            if (rows) *rows = t.rows;
            if (cols) *cols = t.cols;
            return 0;
        }
        i -= (int)nd.tensors.size();
        base += nd.n_params;
    }
    return tvc::set_error(TVC_EINVAL, "tensor index %d out of range", idx);
}

int tvc_sac_create(const tvc_sac_cfg* cfg, int32_t device, float* params, float* grads, float* adam_m, float* adam_v,
                   const float* pe_host, tvc_sac** out) {
    if (!out) return tvc::set_error(TVC_EINVAL, "out is NULL");
    *out = nullptr;
    if (int e = validate(cfg)) return e;
    if (!params || !grads || !adam_m || !adam_v) return tvc::set_error(TVC_EINVAL, "params/grads/adam buffers must be non-NULL");
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(adam_m) |
         reinterpret_cast<uintptr_t>(adam_v)) & 15)
        return tvc::set_error(TVC_EINVAL, "params/grads/adam buffers must be 16-byte aligned");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return tvc::set_error(TVC_ENODEV, "no HIP device visible: libtvc_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return tvc::set_error(TVC_EINVAL, "device %d out of range", device);
    TVC_HIP_CHECK(hipSetDevice(device));
    tvc_sac* h = new (std::nothrow) tvc_sac();
    if (!h) return tvc::set_error(TVC_ENOMEM, "host allocation failed");
    h->cfg = *cfg; h->device = device;
    h->actor = build_actor(*cfg); h->critic = build_critic(*cfg);
    h->actor_inf = derive_infer(h->actor, h->fold, cfg->pe_rows == 1);
    h->n_actor = h->actor.n_params; h->n_critic = h->critic.n_params;
    h->params = params; h->grads = grads; h->adam_m = adam_m; h->adam_v = adam_v;
    const int B = cfg->batch_size, A = cfg->act_dim, NA = cfg->max_act_rows;
    int maxd = 0;
    for (size_t b = 1; b < h->actor.buf_dim.size(); ++b) maxd = std::max(maxd, h->actor.buf_dim[b]);
    long bytes = ctx_bytes(h->actor, 2 * B, 1, true) + ctx_bytes(h->critic, B, 4, true) + 8L * NA * maxd * 4 + 2L * B * cfg->obs_dim * 4 + 512;
    bytes += (long)cfg->pe_rows * cfg->d_model * 4 + (long)B * (cfg->obs_dim + A) * 4 * 2 + (long)B * 64 + (1 << 16);
    bytes += 256L * (4 * (h->actor.buf_dim.size() + h->critic.buf_dim.size()) * 3 + 64);
    h->ov_floats = (long)std::max(1, h->fold.layers) * ((long)h->fold.d * h->fold.d + h->fold.d + 4) +
                   (long)h->fold.d * (h->fold.obs + 1) + 8;
    bytes += 2 * h->ov_floats * 4 + h->n_actor * 4 + 2048;
    const bool want_dctx = cfg->family == 0 && cfg->dropout_p > 0.0f;
    if (want_dctx) bytes += 8L * NA * maxd * 4 + 4096;
    std::vector<PackTile> ptiles;
    std::vector<PackVec> pvecs;
    h->rows_ok = rows_supported(*cfg, h->fold);
    if (h->rows_ok) {
        rows_tables(*cfg, h->actor, h->fold, ptiles, pvecs);
        const char* hn[4] = {"policy_head.6.weight", "policy_head.6.bias", "policy_head.8.weight", "policy_head.8.bias"};
        for (int k = 0; k < 4; ++k) {
            h->head_off[k] = -1;
            for (const TensorInfo& t : h->actor.tensors)
                if (t.name == hn[k]) h->head_off[k] = t.off;
            h->rows_ok = h->rows_ok && h->head_off[k] >= 0;
        }
        for (const PackVec& v : pvecs) h->rows_ok = h->rows_ok && v.src >= 0;
        for (const PackTile& t : ptiles) h->rows_ok = h->rows_ok && t.src >= 0;
        h->rows_tiles = (int)ptiles.size(); h->rows_vecs = (int)pvecs.size();
        h->pack_floats = (long)h->rows_tiles * 4096 + (long)cfg->n_layers * AR_LAYER_VEC + AR_TAIL_VEC + (cfg->use_se ? AR_SE_VEC : 0);
        bytes += 2 * h->pack_floats * 4 + ptiles.size() * sizeof(PackTile) + pvecs.size() * sizeof(PackVec) + 2048;
    }
    std::vector<PackTile3> xtiles;
    if (h->rows_ok) {
        rows_tables_x3(*cfg, h->actor, h->fold, xtiles);
        for (const PackTile3& t : xtiles) h->rows_ok = h->rows_ok && t.src >= 0;
        h->x3_tiles = (int)xtiles.size();
        bytes += 2L * h->x3_tiles * X3_TILE_BYTES + xtiles.size() * sizeof(PackTile3) + 2048;
    }
    std::vector<PackTile> tptiles;
    std::vector<PackVec> tpvecs;
    h->train_rows_ok = h->rows_ok && want_dctx && !cfg->use_se;
    if (h->train_rows_ok) {
        rows_tables_train(*cfg, h->actor, tptiles, tpvecs);
        for (const PackVec& v : tpvecs) h->train_rows_ok = h->train_rows_ok && v.src >= 0;
        for (const PackTile& t : tptiles) h->train_rows_ok = h->train_rows_ok && t.src >= 0;
        h->trows_tiles = (int)tptiles.size(); h->trows_vecs = (int)tpvecs.size();
        h->tpack_floats = (long)h->trows_tiles * 4096 + train_vec_floats(*cfg);
        bytes += 2 * h->tpack_floats * 4 + tptiles.size() * sizeof(PackTile) + tpvecs.size() * sizeof(PackVec) + 2048;
    }
    std::vector<PackTile3> xttiles;
    if (h->train_rows_ok) {
        rows_tables_x3_train(*cfg, h->actor, xttiles);
        bool ok = true;
        for (const PackTile3& t : xttiles) ok = ok && t.src >= 0;
        if (ok) {
            h->x3t_tiles = (int)xttiles.size();
            bytes += 2L * h->x3t_tiles * X3_TILE_BYTES + xttiles.size() * sizeof(PackTile3) + 2048;
        }
    }
    hipError_t he = hipMalloc(&h->slab, bytes);
    if (he != hipSuccess) {
        delete h;
        return tvc::set_error(TVC_ENOMEM, "hipMalloc(%ld bytes) failed: %s", bytes, hipGetErrorString(he));
    }
    (void)hipMemset(h->slab, 0, bytes);
    char* p = (char*)h->slab;
    ctx_alloc_train(h->actx, h->actor, 2 * B, 1, p);
    ctx_alloc_train(h->cctx, h->critic, B, 4, p);  // groups 0, 1 = the online critics (saved for the backward), 2, 3 = the targets
    if (ctx_alloc_infer(h->ictx, h->actor_inf, NA, p) != 0) {
        (void)hipFree(h->slab);
        delete h;
        return tvc::set_error(TVC_EINVAL, "inference slot allocation failed");
    }
    if (want_dctx) {
        h->dctx_ok = ctx_alloc_infer(h->dctx, h->actor, NA, p) == 0;
        h->act_ctr = (int*)carve(p, 16);
    }
    h->pe = (float*)carve(p, (long)cfg->pe_rows * cfg->d_model * 4);
    h->xcat = (float*)carve(p, (long)B * (cfg->obs_dim + A) * 4);
    h->xs2 = (float*)carve(p, 2L * B * cfg->obs_dim * 4);
    h->acat = (float*)carve(p, 2L * B * A * 4);
    h->a_tmp = h->acat + (long)B * A;
    h->ls_tmp = (float*)carve(p, (long)B * A * 4);
    h->mean_tmp = (float*)carve(p, (long)B * A * 4);
    h->y = (float*)carve(p, (long)B * 4);
    h->dq = (float*)carve(p, (long)B * 2 * 4);
    h->tq = (float*)carve(p, (long)B * 2 * 4);
    h->clk = (AdamClock*)carve(p, 2 * sizeof(AdamClock));
    h->ov = (float*)carve(p, h->ov_floats * 4);
    h->snap_ov = (float*)carve(p, h->ov_floats * 4);
    h->snap_p = (float*)carve(p, h->n_actor * 4);
    if (h->rows_ok) {
        h->pack = (float*)carve(p, h->pack_floats * 4);
        h->snap_pack = (float*)carve(p, h->pack_floats * 4);
        h->d_ptiles = (PackTile*)carve(p, ptiles.size() * sizeof(PackTile));
        h->d_pvecs = (PackVec*)carve(p, pvecs.size() * sizeof(PackVec));
        h->xpack = carve(p, (long)h->x3_tiles * X3_TILE_BYTES);
        h->snap_xpack = carve(p, (long)h->x3_tiles * X3_TILE_BYTES);
        h->d_xtiles = (PackTile3*)carve(p, xtiles.size() * sizeof(PackTile3));
    }
    if (h->train_rows_ok) {
        h->tpack = (float*)carve(p, h->tpack_floats * 4);
        h->snap_tpack = (float*)carve(p, h->tpack_floats * 4);
        h->d_tptiles = (PackTile*)carve(p, tptiles.size() * sizeof(PackTile));
        h->d_tpvecs = (PackVec*)carve(p, tpvecs.size() * sizeof(PackVec));
        if (h->x3t_tiles > 0) {
            h->xtpack = carve(p, (long)h->x3t_tiles * X3_TILE_BYTES);
            h->snap_xtpack = carve(p, (long)h->x3t_tiles * X3_TILE_BYTES);
            h->d_xttiles = (PackTile3*)carve(p, xttiles.size() * sizeof(PackTile3));
        }
    }
    if ((long)(p - (char*)h->slab) > bytes) {
        (void)hipFree(h->slab);
        delete h;
        return tvc::set_error(TVC_ENOMEM, "internal: workspace under-sized");
    }
    if (cfg->family == 0) {
        if (!pe_host) {
            (void)hipFree(h->slab);
            delete h;
            return tvc::set_error(TVC_EINVAL, "family 0 needs the positional-encoding table");
        }
        he = hipMemcpy(h->pe, pe_host, (long)cfg->pe_rows * cfg->d_model * 4, hipMemcpyHostToDevice);
        if (he != hipSuccess) {
            (void)hipFree(h->slab);
            delete h;
            return tvc::set_error(TVC_EHIP, "hipMemcpy(pe) failed: %s", hipGetErrorString(he));
        }
    }
    if (h->rows_ok) {
        he = hipMemcpy(h->d_ptiles, ptiles.data(), ptiles.size() * sizeof(PackTile), hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemcpy(h->d_pvecs, pvecs.data(), pvecs.size() * sizeof(PackVec), hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemcpy(h->d_xtiles, xtiles.data(), xtiles.size() * sizeof(PackTile3), hipMemcpyHostToDevice);
        if (he == hipSuccess && h->train_rows_ok) {
            he = hipMemcpy(h->d_tptiles, tptiles.data(), tptiles.size() * sizeof(PackTile), hipMemcpyHostToDevice);
            if (he == hipSuccess) he = hipMemcpy(h->d_tpvecs, tpvecs.data(), tpvecs.size() * sizeof(PackVec), hipMemcpyHostToDevice);
            if (he == hipSuccess && h->x3t_tiles > 0)
                he = hipMemcpy(h->d_xttiles, xttiles.data(), xttiles.size() * sizeof(PackTile3), hipMemcpyHostToDevice);
        }
        if (he != hipSuccess) {
            (void)hipFree(h->slab);
            delete h;
            return tvc::set_error(TVC_EHIP, "hipMemcpy(pack tables) failed: %s", hipGetErrorString(he));
        }
    }
    {
        AdamClock c0[2] = {{1.0, 1.0, 0, 0}, {1.0, 1.0, 0, 0}};
        he = hipMemcpy(h->clk, c0, sizeof(c0), hipMemcpyHostToDevice);
        if (he != hipSuccess) {
            (void)hipFree(h->slab);
            delete h;
            return tvc::set_error(TVC_EHIP, "hipMemcpy(adam clocks) failed: %s", hipGetErrorString(he));
        }
    }
    *out = h;
    return 0;
}

void tvc_sac_destroy(tvc_sac* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->slab);
    delete h;
}

static const float* P_head(tvc_sac* h, int k) { return h->P_actor() + h->head_off[k]; }

// Split-operand acting stream (tvc_actor_x3.h): packed from the folded weights the last update left, then re-packed by every policy
// update and copied by every snapshot.  Idempotent; stream-ordered (call it where no update runs on another stream).
static int x3_make_live(tvc_sac* h, hipStream_t st) {
    if (h->x3_live) return 0;
    if (!h->rows_ok) return tvc::set_error(TVC_EINVAL, "split-operand acting needs the one-launch acting path (family 0 reference shapes)");
    h->x3_live = true;
    const float* P = h->P_actor();
    HeadPack hp{P + h->head_off[0], P + h->head_off[1], P + h->head_off[2], P + h->head_off[3], 2 * h->cfg.act_dim,
                h->pack + (long)h->rows_tiles * 4096 + (long)h->cfg.n_layers * AR_LAYER_VEC, nullptr};
    PackSet none{nullptr, 0, nullptr, 0, nullptr, nullptr};
    hipLaunchKernelGGL(pack_actor_kernel, dim3(1 + h->x3_tiles), dim3(256), 0, st, P, h->ov, none, none, hp,
                       Ticks{nullptr, 0.f, 0.f, nullptr}, PackSet3{h->d_xtiles, h->x3_tiles, h->xpack}, PackSet3{nullptr, 0, nullptr});
    // (a snapshot taken before this stream existed holds the same parameters unless an update ran in between)
    TVC_HIP_CHECK(hipMemcpyAsync(h->snap_xpack, h->xpack, (size_t)h->x3_tiles * X3_TILE_BYTES, hipMemcpyDeviceToDevice, st));
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}
int tvc_sac_enable_x3(tvc_sac* h, void* stream) {
    if (!h) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    return x3_make_live(h, (hipStream_t)stream);
}

int tvc_sac_act(tvc_sac* h, const float* obs, int32_t n, const float* eps, float* act, float* mean, float* logstd, int32_t flags,
                void* stream) {
    if (!h || !obs || !act) return tvc::set_error(TVC_EINVAL, "null argument");
    if (n < 1 || n > h->cfg.max_act_rows) return tvc::set_error(TVC_EINVAL, "n=%d outside [1, max_act_rows=%d]", n, h->cfg.max_act_rows);
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int A = h->cfg.act_dim;
    // activation group strides depend on the row count actually used
    for (size_t b = 1; b < h->ictx.gY.size(); ++b) h->ictx.gY[b] = (long)n * h->actor_inf.buf_dim[b];
    const bool snap = (flags & 2) != 0;  // read the snapshot taken by tvc_sac_snapshot_policy instead of the live parameters
    if (flags & 8) {
        // acting in TRAIN mode, like the reference, which never calls .eval() (agent/...:765): the net as trained (attention not
        // folded -- the attention-weight dropout zeroes whole heads of V) with fresh masks at every dropout site per call
        if (!h->dctx_ok) return tvc::set_error(TVC_EINVAL, "train-mode acting needs family 0 with dropout_p > 0");
        if (h->train_rows_ok && n >= split_min_rows() && g_force_variant == 0) {
            // one launch: the split kernel's TRAIN instantiation on the stream of the net as trained (tvc_actor_split.h)
            if (!h->train_stream_live) {  // first request: pack that stream now; from here on every policy update re-packs it
                h->train_stream_live = true;
                const float* P = h->P_actor();
                HeadPack hp{P + h->head_off[0], P + h->head_off[1], P + h->head_off[2], P + h->head_off[3], 2 * A,
                            h->pack + (long)h->rows_tiles * 4096 + (long)h->cfg.n_layers * AR_LAYER_VEC,
                            h->tpack + (long)h->trows_tiles * 4096 + 256 + 512L * h->cfg.n_layers};
                PackSet none{nullptr, 0, nullptr, 0, nullptr, nullptr};
                PackSet s1{h->d_tptiles, h->trows_tiles, h->d_tpvecs, h->trows_vecs, reinterpret_cast<float4*>(h->tpack),
                           h->tpack + (long)h->trows_tiles * 4096};
                hipLaunchKernelGGL(pack_actor_kernel, dim3(s1.n_tiles + s1.n_vecs + 1), dim3(256), 0, st, P, h->ov, none, s1, hp,
                                   Ticks{nullptr, 0.f, 0.f, nullptr}, PackSet3{nullptr, 0, nullptr}, PackSet3{nullptr, 0, nullptr});
                // (the snapshot, if one is being read, was taken before this stream existed: it holds the same parameters)
                TVC_HIP_CHECK(hipMemcpyAsync(h->snap_tpack, h->tpack, h->tpack_floats * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
            const float* pk = snap ? h->snap_pack : h->pack;
            const float* tpk = snap ? h->snap_tpack : h->tpack;
            ActRowsArgs a{};
            a.obs = obs; a.eps = eps; a.act = act; a.mean = mean; a.logstd = logstd;
            a.tiles = reinterpret_cast<const float4*>(tpk); a.vec = pk + (long)h->rows_tiles * 4096;
            a.tvec = tpk + (long)h->trows_tiles * 4096; a.pe0 = h->pe;
            a.M = n; a.obs_dim = h->cfg.obs_dim; a.A = A; a.clamp_act = (flags & 1) ? 0 : 1;
            a.n_layers = h->cfg.n_layers; a.n_tiles = h->trows_tiles; a.stamps = nullptr; a.use_se = 0;
            a.drop_ctr = h->act_ctr; a.drop_thresh = (unsigned)lroundf(h->cfg.dropout_p * 65536.0f);
            a.drop_scale = 65536.0f / (float)(65536u - a.drop_thresh); a.drop_seed = h->cfg.dropout_seed;
            size_t dyn_lds = 0;
            if ((flags & 16) && h->x3t_tiles > 0 && n >= rows_min_rows()) {
                // ... from 16 384 rows with bit 4: the row-owner kernel on the bf16 matrix pipe (actor_x3_kernel<true>) on the same net's
                // split-operand stream; the vector sections are the ones above
                if (!h->x3t_live) {
                    h->x3t_live = true;
                    PackSet none{nullptr, 0, nullptr, 0, nullptr, nullptr};
                    HeadPack hp{P_head(h, 0), P_head(h, 1), P_head(h, 2), P_head(h, 3), 2 * A,
                                h->pack + (long)h->rows_tiles * 4096 + (long)h->cfg.n_layers * AR_LAYER_VEC,
                                h->tpack + (long)h->trows_tiles * 4096 + 256 + 512L * h->cfg.n_layers};
                    hipLaunchKernelGGL(pack_actor_kernel, dim3(1 + h->x3t_tiles), dim3(256), 0, st, h->P_actor(), h->ov, none, none, hp,
                                       Ticks{nullptr, 0.f, 0.f, nullptr}, PackSet3{nullptr, 0, nullptr},
                                       PackSet3{h->d_xttiles, h->x3t_tiles, h->xtpack});
                    TVC_HIP_CHECK(hipMemcpyAsync(h->snap_xtpack, h->xtpack, (size_t)h->x3t_tiles * X3_TILE_BYTES, hipMemcpyDeviceToDevice, st));
                }
                a.tiles = reinterpret_cast<const float4*>(snap ? h->snap_xtpack : h->xtpack);
                a.n_tiles = h->x3t_tiles;
                if (flags & 4) {
                    if (!h->x3t_attr_set) {
                        TVC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(actor_x3_kernel<true>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
                        h->x3t_attr_set = true;
                    }
                    dyn_lds = 65536;
                }
                hipLaunchKernelGGL(actor_x3_kernel<true>, dim3((n + 16 * X3_NW - 1) / (16 * X3_NW)), dim3(64 * X3_NW), dyn_lds, st, a);
                hipLaunchKernelGGL(act_ctr_tick_kernel, dim3(1), dim3(1), 0, st, h->act_ctr);  // every workgroup has read the counter
                TVC_HIP_CHECK(hipGetLastError());
                return 0;
            }
            if (flags & 4) {
                if (!h->tsplit_attr_set) {
                    TVC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(actor_split_kernel<true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
                    h->tsplit_attr_set = true;
                }
                dyn_lds = 32768;
            }
            hipLaunchKernelGGL(actor_split_kernel<true>, dim3((n + 15) / 16), dim3(256), dyn_lds, st, a);
            hipLaunchKernelGGL(act_ctr_tick_kernel, dim3(1), dim3(1), 0, st, h->act_ctr);  // every workgroup has read the counter
            TVC_HIP_CHECK(hipGetLastError());
            return 0;
        }
        for (size_t b = 1; b < h->dctx.gY.size(); ++b) h->dctx.gY[b] = (long)n * h->actor.buf_dim[b];
        DropCtl dc;
        dc.ctr = h->act_ctr; dc.thresh = (unsigned)lroundf(h->cfg.dropout_p * 65536.0f);
        dc.scale = 65536.0f / (float)(65536u - dc.thresh); dc.site_base = 300; dc.seed = h->cfg.dropout_seed;
        net_forward(h->actor, snap ? h->snap_p : h->P_actor(), 0, obs, 0, n, 1, h->dctx, false, h->pe, h->cfg.pe_rows, st, nullptr,
                    nullptr, &dc);
        hipLaunchKernelGGL(sample_action_kernel, dim3((n * A + 255) / 256), dim3(256), 0, st, h->dctx.Y.back(), eps, act, mean, logstd,
                           n, A, (flags & 1) ? 0 : 1, Ticks{nullptr, 0.f, 0.f, h->act_ctr});  // ... and advances the call counter
        TVC_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if ((flags & 16) && h->rows_ok && n >= rows_min_rows() && g_force_variant == 0) {
        // the whole pass as one launch on the bf16 matrix pipe, operands split in three bf16 terms, fp32-exact (tvc_actor_x3.h)
        if (int e = x3_make_live(h, st)) return e;
        const float* pk = snap ? h->snap_pack : h->pack;
        ActRowsArgs a{};
        a.obs = obs; a.eps = eps; a.act = act; a.mean = mean; a.logstd = logstd;
        a.tiles = reinterpret_cast<const float4*>(snap ? h->snap_xpack : h->xpack); a.vec = pk + (long)h->rows_tiles * 4096;
        a.M = n; a.obs_dim = h->cfg.obs_dim; a.A = A; a.clamp_act = (flags & 1) ? 0 : 1;
        a.n_layers = h->cfg.n_layers; a.n_tiles = h->x3_tiles; a.stamps = h->rows_stamps; a.use_se = h->cfg.use_se;
        size_t dyn_lds = 0;
        if (flags & 4) {  // "share the CUs": 48 KB static + 64 KB unused dynamic LDS -> one workgroup per CU instead of two
            if (!h->x3_attr_set) {
                TVC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(actor_x3_kernel<false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
                h->x3_attr_set = true;
            }
            dyn_lds = 65536;
        }
        hipLaunchKernelGGL(actor_x3_kernel<false>, dim3((n + 16 * X3_NW - 1) / (16 * X3_NW)), dim3(64 * X3_NW), dyn_lds, st, a);
        TVC_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (h->rows_ok && n >= rows_min_rows() && g_force_variant == 0) {  // the whole pass as one launch (tvc_actor_rows.h)
        const float* pk = snap ? h->snap_pack : h->pack;
        ActRowsArgs a{};
        a.obs = obs; a.eps = eps; a.act = act; a.mean = mean; a.logstd = logstd;
        a.tiles = reinterpret_cast<const float4*>(pk); a.vec = pk + (long)h->rows_tiles * 4096;
        a.M = n; a.obs_dim = h->cfg.obs_dim; a.A = A; a.clamp_act = (flags & 1) ? 0 : 1;
        a.n_layers = h->cfg.n_layers; a.n_tiles = h->rows_tiles; a.stamps = h->rows_stamps; a.use_se = h->cfg.use_se;
        // flags bit 2 ("share the CUs"): 64 KB of unused dynamic LDS make the kernel fit once per CU instead of twice, which
        // leaves half of every CU's registers (and 64 KB of LDS) to whatever runs on other streams -- the ~100 small kernels of
        // a SAC update beside the acting pass.  Two workgroups per CU own every VGPR of the CU: faster alone (1.9 vs 2.4 ms at
        // 65 536 rows), but nothing else can then run until they retire.
        size_t dyn_lds = 0;
        if (flags & 4) {
            if (!h->lds_attr_set) {  // per device (hipSetDevice above), remembered per handle
                TVC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(actor_rows_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
                h->lds_attr_set = true;
            }
            dyn_lds = 65536;
        }
        hipLaunchKernelGGL(actor_rows_kernel, dim3((n + 16 * AR_NW - 1) / (16 * AR_NW)), dim3(64 * AR_NW), dyn_lds, st, a);
        TVC_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (h->rows_ok && !h->cfg.use_se && n >= split_min_rows() && g_force_variant == 0) {
        // the whole pass as one launch, 16 rows per workgroup, the four waves splitting every Linear (tvc_actor_split.h)
        const float* pk = snap ? h->snap_pack : h->pack;
        ActRowsArgs a{};
        a.obs = obs; a.eps = eps; a.act = act; a.mean = mean; a.logstd = logstd;
        a.tiles = reinterpret_cast<const float4*>(pk); a.vec = pk + (long)h->rows_tiles * 4096;
        a.M = n; a.obs_dim = h->cfg.obs_dim; a.A = A; a.clamp_act = (flags & 1) ? 0 : 1;
        a.n_layers = h->cfg.n_layers; a.n_tiles = h->rows_tiles; a.stamps = nullptr; a.use_se = 0;
        size_t dyn_lds = 0;
        if (flags & 4) {  // "share the CUs": 32 KB of unused dynamic LDS -> one workgroup (4 waves) per CU instead of two
            if (!h->split_attr_set) {
                TVC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(actor_split_kernel<false>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
                h->split_attr_set = true;
            }
            dyn_lds = 32768;
        }
        hipLaunchKernelGGL(actor_split_kernel<false>, dim3((n + 15) / 16), dim3(256), dyn_lds, st, a);
        TVC_HIP_CHECK(hipGetLastError());
        return 0;
    }
    net_forward(h->actor_inf, snap ? h->snap_p : h->P_actor(), 0, obs, 0, n, 1, h->ictx, false, h->cfg.family == 0 ? h->pe : nullptr,
                h->cfg.pe_rows, st, snap ? h->snap_ov : h->ov);
    const float* head = h->ictx.Y.back();
    hipLaunchKernelGGL(sample_action_kernel, dim3((n * A + 255) / 256), dim3(256), 0, st, head, eps, act, mean, logstd, n, A,
                       (flags & 1) ? 0 : 1, Ticks{nullptr, 0.f, 0.f, nullptr});
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

// Diagnostics: in-kernel shader clock of the acting megakernel (guide: DVFS give-back item 6): Delta s_memtime / Delta
// s_memrealtime x 100 MHz per workgroup, median over workgroups of the last of `launches` back-to-back launches.
// out[0] = clock MHz, out[1] = median workgroup lifetime in us, out[2] = workgroups.  Synchronises.
static int rows_probe(tvc_sac* h, const float* obs, int32_t n, int32_t launches, int32_t flags, std::vector<unsigned long long>& v,
                      void* stream) {
    if (!h || !obs || n < 1 || launches < 1) return tvc::set_error(TVC_EINVAL, "bad argument");
    if (!h->rows_ok) return tvc::set_error(TVC_EINVAL, "this handle does not use the row-owner acting kernel");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    const int nwg = (n + 16 * AR_NW - 1) / (16 * AR_NW), A = h->cfg.act_dim;
    unsigned long long* st = nullptr;
    float* act = nullptr;
    TVC_HIP_CHECK(hipMalloc((void**)&st, (size_t)nwg * AR_STAMPS * sizeof(unsigned long long)));
    if (hipMalloc((void**)&act, (size_t)n * A * sizeof(float)) != hipSuccess) { (void)hipFree(st); return tvc::set_error(TVC_ENOMEM, "hipMalloc failed"); }
    h->rows_stamps = st;
    int rc = 0;
    for (int i = 0; i < launches && rc == 0; ++i) rc = tvc_sac_act(h, obs, n, nullptr, act, nullptr, nullptr, flags & (4 | 16), stream);
    h->rows_stamps = nullptr;
    hipError_t he = hipStreamSynchronize((hipStream_t)stream);
    v.assign((size_t)nwg * AR_STAMPS, 0);
    if (rc == 0 && he == hipSuccess) he = hipMemcpy(v.data(), st, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(st); (void)hipFree(act);
    if (rc) return rc;
    if (he != hipSuccess) return tvc::set_error(TVC_EHIP, "rows clock probe failed: %s", hipGetErrorString(he));
    return 0;
}
int tvc_debug_rows_clock(tvc_sac* h, const float* obs, int32_t n, int32_t launches, double* out, void* stream) {
    if (!out) return tvc::set_error(TVC_EINVAL, "bad argument");
    std::vector<unsigned long long> v;
    if (int rc = rows_probe(h, obs, n, launches, 0, v, stream)) return rc;
    const int nwg = (n + 16 * AR_NW - 1) / (16 * AR_NW);
    std::vector<double> mhz, life;
    for (int b = 0; b < nwg; ++b) {
        const double dc = (double)(v[AR_STAMPS * b + 2] - v[AR_STAMPS * b]), dr = (double)(v[AR_STAMPS * b + 3] - v[AR_STAMPS * b + 1]);
        if (dr > 0) { mhz.push_back(dc / dr * 100.0); life.push_back(dr / 100.0); }
    }
    if (mhz.empty()) return tvc::set_error(TVC_EHIP, "no stamps came back");
    std::sort(mhz.begin(), mhz.end()); std::sort(life.begin(), life.end());
    out[0] = mhz[mhz.size() / 2]; out[1] = life[life.size() / 2]; out[2] = (double)nwg;
    return 0;
}
// ... and the raw stamps of the last launch: 6 x uint64 per workgroup {s_memtime, s_memrealtime (100 MHz) at start, the same at
// the end, XCC_ID, HW_ID} into a HOST buffer of ceil(n / 64) * 6 entries; flags bit 2 as in tvc_sac_act.
int tvc_debug_rows_stamps(tvc_sac* h, const float* obs, int32_t n, int32_t launches, int32_t flags, uint64_t* out_host, void* stream) {
    if (!out_host) return tvc::set_error(TVC_EINVAL, "bad argument");
    std::vector<unsigned long long> v;
    if (int rc = rows_probe(h, obs, n, launches, flags, v, stream)) return rc;
    for (size_t i = 0; i < v.size(); ++i) out_host[i] = v[i];
    return 0;
}

// The critics' clock advance normally rides on the sampling kernel of the actor phase; any other entry point that finds it pending
// issues it as its own (tiny) launch first, so the clock a kernel reads is always exact.
static void flush_tick(tvc_sac* h, hipStream_t st) {
    if (!h->tick_pending) return;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, h->clk + 0, h->cfg.adam_b1, h->cfg.adam_b2);
    h->tick_pending = false;
}

// critic input [s | a]: read in place by the thin first-layer kernels when it is at most 16 wide, else concatenated
static const float* critic_input(tvc_sac* h, const float* s, const float* a, In2& in2, hipStream_t st) {
    const int B = h->cfg.batch_size, A = h->cfg.act_dim, no = h->cfg.obs_dim;
    if (no + A <= THIN_K) {
        in2.X2 = a; in2.gX2 = 0; in2.K1 = no; in2.ldx = no; in2.ldx2 = A;
        return s;
    }
    in2.X2 = nullptr;
    hipLaunchKernelGGL(concat_kernel, dim3((B * (no + A) + 255) / 256), dim3(256), 0, st, s, a, h->xcat, B, no, A);
    return h->xcat;
}

// dropout control of one forward call of an update: site bases 0 (actor), 100 + 20 * call (critics: call 1 = targets,
// 2 = online nets in the critic loss, 3 = online nets in the policy loss); the counter is the actor's Adam step count,
// constant during an update and advanced by its last kernel
static const DropCtl* drop_ctl(tvc_sac* h, DropCtl& dc, unsigned site_base) {
    if (!(h->cfg.dropout_p > 0.0f)) return nullptr;
    const unsigned thresh = (unsigned)lroundf(h->cfg.dropout_p * 65536.0f);
    dc.ctr = &h->clk[1].step; dc.thresh = thresh; dc.scale = 65536.0f / (float)(65536u - thresh); dc.site_base = site_base;
    dc.seed = h->cfg.dropout_seed;
    return &dc;
}

static bool merge_critic_forwards() {  // TVC_MERGE_CRITICS=0: the target and the online critics' forward as two launches chains (A/B)
    static const bool v = [] { const char* e = getenv("TVC_MERGE_CRITICS"); return !e || atoi(e) != 0; }();
    return v;
}
static int check_batch_ptrs(const void* a, const void* b, const void* c) {
    if (!a || !b || !c) return tvc::set_error(TVC_EINVAL, "null batch pointer");
    return 0;
}

int tvc_sac_critic_grads(tvc_sac* h, const float* s, const float* a, const float* r, const float* s2, const float* d,
                         const float* eps_next, float* losses, void* stream) {
    if (!h || !losses) return tvc::set_error(TVC_EINVAL, "null argument");
    if (h->cfg.use_se) return tvc::set_error(TVC_EINVAL, "use_se nets are acting-only (the reference never trains its hierarchical policy)");
    if (check_batch_ptrs(s, a, r) || check_batch_ptrs(s2, d, eps_next)) return TVC_EINVAL;
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    flush_tick(h, st);
    const tvc_sac_cfg& c = h->cfg;
    const int B = c.batch_size, A = c.act_dim, no = c.obs_dim;
    const float* pe = c.family == 0 ? h->pe : nullptr;
    // ONE actor forward over [s ; s'] (2B rows): the policy parameters do not change between the target pass on s'
    // (here) and the policy pass on s (tvc_sac_actor_grads), so the rows of s are computed -- and saved for the
    // backward -- now, and the positional-encoding table is indexed modulo its rows
    const bool one_critic_pass = no + A <= THIN_K && merge_critic_forwards();
    hipLaunchKernelGGL(update_prep_kernel, dim3(1 + (2 * B * no + 255) / 256), dim3(256), 0, st, s, a, s2, losses, h->xs2, B, no, A,
                       0.1f, one_critic_pass ? h->acat : (float*)nullptr);
    DropCtl dca, dcq;
    net_forward(h->actor, h->P_actor(), 0, h->xs2, 0, 2 * B, 1, h->actx, true, pe, c.pe_rows, st, nullptr, nullptr,
                drop_ctl(h, dca, 0));
    h->actor_fwd_valid = true;
    // target: a' ~ pi(s'), y = r + gamma (1-d) min(tq1, tq2)(s', a')
    hipLaunchKernelGGL(sample_action_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, h->actx.Y.back() + (long)B * 2 * A,
                       eps_next, h->a_tmp, (float*)nullptr, (float*)nullptr, B, A, 0, Ticks{nullptr, 0.f, 0.f, nullptr});
    In2 in2;
    const float* x;
    const float* tq = h->tq;
    if (one_critic_pass) {
        // ONE forward over four nets: groups 0, 1 = the online critics on (s, a) (saved for the backward), groups 2, 3 = the target
        // critics on (s', a') -- the target parameters lie behind the online ones, the inputs are the halves of [s ; s'] and [a ; a']
        // (input group = z / 2), the dropout sites stay those of the two separate calls (140 + op for z < 2, 120 + op and z - 2 behind)
        In2 in4;
        in4.X2 = h->acat; in4.gX2 = (long)B * A; in4.K1 = no; in4.ldx = no; in4.ldx2 = A; in4.gdiv = 2;
        const DropCtl* dq = drop_ctl(h, dcq, 140);
        if (dq) { dcq.zsplit = 2; dcq.site_base2 = 120; }
        net_forward(h->critic, h->P_q(), h->n_critic, h->xs2, (long)B * no, B, 4, h->cctx, true, nullptr, 0, st, nullptr, &in4, dq);
        tq = h->cctx.Y.back() + 2 * h->cctx.gY.back();
        x = critic_input(h, s, a, in2, st);  // (what the backward reads: the same values as the first halves above)
    } else {
        x = critic_input(h, s2, h->a_tmp, in2, st);
        {   // the target critics' output goes to its own buffer: the online pass below reuses the context, and the loss kernel forms
            // y = r + gamma (1 - d) min(tq1, tq2) itself
            float* keep = h->cctx.Y.back();
            h->cctx.Y.back() = h->tq;
            net_forward(h->critic, h->P_tq(), h->n_critic, x, 0, B, 2, h->cctx, false, nullptr, 0, st, nullptr, in2.X2 ? &in2 : nullptr,
                        drop_ctl(h, dcq, 120));
            h->cctx.Y.back() = keep;
        }
        // online critics on (s, a): forward (saved), loss, backward
        x = critic_input(h, s, a, in2, st);
        net_forward(h->critic, h->P_q(), h->n_critic, x, 0, B, 2, h->cctx, true, nullptr, 0, st, nullptr, in2.X2 ? &in2 : nullptr,
                    drop_ctl(h, dcq, 140));
    }
    if (one_critic_pass && dcq.zsplit) { dcq.zsplit = 0; dcq.site_base2 = 0; }  // (the backward below regenerates the online nets' masks)
    hipLaunchKernelGGL(q_loss_kernel, dim3((B + 255) / 256, 2), dim3(256), 0, st, h->cctx.Y.back(), tq, r, d, c.gamma,
                       h->cctx.dY.back(), losses, B);
    if (!h->grads_clean[0]) TVC_HIP_CHECK(hipMemsetAsync(h->G_q(), 0, 2 * h->n_critic * sizeof(float), st));
    h->grads_clean[0] = false;
    net_backward(h->critic, h->P_q(), h->n_critic, h->G_q(), h->n_critic, x, 0, B, 2, h->cctx, false, st, in2.X2 ? &in2 : nullptr,
                 drop_ctl(h, dcq, 140));
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

// one launch: Adam over [off, off + n) of the trainable buffer (+ the Polyak update of the targets as extra workgroups when asked);
// the clock it reads is advanced by a later launch of the chain (see AdamClock)
static void adam_apply(tvc_sac* h, float* p, float* g, long off, long n, int which, float gscale, bool polyak, hipStream_t st) {
    const tvc_sac_cfg& c = h->cfg;
    const int blocks = (int)std::min<long>((n / 4 + 255) / 256, 2048);
    const long pn = polyak ? 2 * h->n_critic : 0;
    const int pblocks = polyak ? (int)std::min<long>((pn + 255) / 256, 512) : 0;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks + pblocks), dim3(256), 0, st, p, g, h->adam_m + off, h->adam_v + off, n, c.lr,
                       c.adam_b1, c.adam_b2, c.adam_eps, h->clk + which, gscale, blocks, h->P_tq(), h->P_q(), pn, c.tau);
    h->grads_clean[which] = true;
}

int tvc_sac_critic_apply(tvc_sac* h, float grad_scale, void* stream) {
    if (!h) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    flush_tick(h, (hipStream_t)stream);
    adam_apply(h, h->P_q(), h->G_q(), h->n_actor, 2 * h->n_critic, 0, grad_scale, false, (hipStream_t)stream);
    h->tick_pending = true;  // rides on the first launch of tvc_sac_actor_grads
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_sac_actor_grads(tvc_sac* h, const float* s, const float* eps_new, float* losses, void* stream) {
    if (!h || !s || !eps_new || !losses) return tvc::set_error(TVC_EINVAL, "null argument");
    if (h->cfg.use_se) return tvc::set_error(TVC_EINVAL, "use_se nets are acting-only (the reference never trains its hierarchical policy)");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const tvc_sac_cfg& c = h->cfg;
    const int B = c.batch_size, A = c.act_dim, no = c.obs_dim, nin = no + A;
    const float* pe = c.family == 0 ? h->pe : nullptr;
    const float* xin = s;
    DropCtl dca, dcq;
    if (h->actor_fwd_valid) {  // rows [0, B) of the stacked forward made by tvc_sac_critic_grads of this update
        xin = h->xs2;
        h->actor_fwd_valid = false;
    } else {
        net_forward(h->actor, h->P_actor(), 0, s, 0, B, 1, h->actx, true, pe, c.pe_rows, st, nullptr, nullptr, drop_ctl(h, dca, 0));
    }
    const float* head = h->actx.Y.back();
    hipLaunchKernelGGL(sample_action_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, head, eps_new, h->a_tmp, h->mean_tmp,
                       h->ls_tmp, B, A, 0, Ticks{h->tick_pending ? h->clk + 0 : nullptr, c.adam_b1, c.adam_b2, nullptr});
    h->tick_pending = false;
    In2 in2;
    const float* x = critic_input(h, s, h->a_tmp, in2, st);
    net_forward(h->critic, h->P_q(), h->n_critic, x, 0, B, 2, h->cctx, true, nullptr, 0, st, nullptr, in2.X2 ? &in2 : nullptr,
                drop_ctl(h, dcq, 160));
    hipLaunchKernelGGL(actor_loss_kernel, dim3((B + 255) / 256), dim3(256), 0, st, h->cctx.Y.back(), h->ls_tmp, eps_new,
                       h->cctx.dY.back(), losses, B, A, c.alpha);
    // data gradients only through the critics (the reference also fills q.grad here, then discards it)
    HeadGradFuse hg{head, eps_new, h->actx.dY.back(), A, no, c.alpha, false};
    net_backward(h->critic, h->P_q(), h->n_critic, nullptr, 0, x, 0, B, 2, h->cctx, true, st, in2.X2 ? &in2 : nullptr,
                 drop_ctl(h, dcq, 160), &hg);
    if (!hg.done)  // (first critic layer not on the thin path: separate input-gradient and head-gradient launches)
        hipLaunchKernelGGL(actor_head_grad_kernel, dim3((B * A + 255) / 256), dim3(256), 0, st, h->cctx.dY[0], h->cctx.gY[0], nin, no,
                           head, eps_new, h->actx.dY.back(), B, A, c.alpha);
    if (!h->grads_clean[1]) TVC_HIP_CHECK(hipMemsetAsync(h->G_actor(), 0, h->n_actor * sizeof(float), st));
    h->grads_clean[1] = false;
    net_backward(h->actor, h->P_actor(), 0, h->G_actor(), 0, xin, 0, B, 1, h->actx, false, st, nullptr, drop_ctl(h, dca, 0));
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

// W_ov = W_o W_v (one grouped GEMM launch), then b_ov = W_o b_v + b_o of every layer and W', b' of the folded embedding (one
// launch), then the acting kernel's weight stream + the folded output head (one launch).  `tick`: the actor's Adam clock
// advances in the LAST launch issued here (or in its own launch when there is nothing to derive).
static void refresh_folded(tvc_sac* h, hipStream_t st, AdamClock* tick) {
    const FoldInfo& f = h->fold;
    const tvc_sac_cfg& c = h->cfg;
    if (f.layers == 0) {
        if (tick) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, tick, c.adam_b1, c.adam_b2);
        return;
    }
    const int d = f.d;
    const long ostride = (long)d * d + d;
    GemmArgs g{};  // C[o, k] = sum_j Wo[o, j] Wv[j, k]
    g.A = h->P_actor() + f.o_w; g.B = h->P_actor() + f.v_w; g.C = h->ov;
    g.M = d; g.N = d; g.K = d; g.K1 = d; g.lda = d; g.ldb = d; g.ldc = d;
    g.gA = f.layer_stride; g.gB = f.layer_stride; g.gC = ostride;
    launch_gemm(true, false, g, f.layers, st);
    const Ticks none{nullptr, 0.f, 0.f, nullptr}, tk{tick, c.adam_b1, c.adam_b2, nullptr};
    FoldArgs fa{};
    fa.P = h->P_actor(); fa.OV = h->ov; fa.pe0 = c.family == 0 ? h->pe : nullptr;
    fa.o_w = f.o_w; fa.o_b = f.o_b; fa.v_b = f.v_b; fa.layer_stride = f.layer_stride; fa.ostride = ostride;
    fa.embed = f.embed ? 1 : 0; fa.e_w = f.e_w; fa.e_b = f.e_b; fa.e_off = f.e_off; fa.d = d; fa.obs = f.obs;
    hipLaunchKernelGGL(fold_kernel, dim3(d, f.layers), dim3(256), 0, st, fa, h->rows_ok ? none : tk);
    if (h->rows_ok) {  // re-pack the acting megakernel's weight stream from the fresh parameters / folded weights
        const float* P = h->P_actor();
        HeadPack hp{P + h->head_off[0], P + h->head_off[1], P + h->head_off[2], P + h->head_off[3], 2 * c.act_dim,
                    h->pack + (long)h->rows_tiles * 4096 + (long)c.n_layers * AR_LAYER_VEC,
                    h->train_rows_ok ? h->tpack + (long)h->trows_tiles * 4096 + 256 + 512L * c.n_layers : nullptr};
        PackSet s0{h->d_ptiles, h->rows_tiles, h->d_pvecs, h->rows_vecs, reinterpret_cast<float4*>(h->pack),
                   h->pack + (long)h->rows_tiles * 4096};
        const bool tl = h->train_rows_ok && h->train_stream_live;
        if (!tl) hp.tail_t = nullptr;
        PackSet s1{h->d_tptiles, tl ? h->trows_tiles : 0, h->d_tpvecs, tl ? h->trows_vecs : 0,
                   reinterpret_cast<float4*>(h->tpack), tl ? h->tpack + (long)h->trows_tiles * 4096 : nullptr};
        const PackSet3 s3{h->d_xtiles, h->x3_live ? h->x3_tiles : 0, h->xpack};
        const PackSet3 s4{h->d_xttiles, h->x3t_live ? h->x3t_tiles : 0, h->xtpack};
        hipLaunchKernelGGL(pack_actor_kernel, dim3(s0.n_tiles + s0.n_vecs + s1.n_tiles + s1.n_vecs + 1 + s3.n_tiles + s4.n_tiles), dim3(256),
                           0, st, h->P_actor(), h->ov, s0, s1, hp, tk, s3, s4);
    }
}

// Adam step counters (critics, actor) live on the device so that a captured update keeps counting; these two calls
// synchronise and are meant for checkpoints only
static int set_clocks(tvc_sac* h, const int32_t steps[2]) {
    AdamClock c[2];
    for (int i = 0; i < 2; ++i) {
        c[i].step = steps[i]; c[i].pad = 0;
        c[i].b1t = pow((double)h->cfg.adam_b1, (double)steps[i]);
        c[i].b2t = pow((double)h->cfg.adam_b2, (double)steps[i]);
    }
    TVC_HIP_CHECK(hipMemcpy(h->clk, c, sizeof(c), hipMemcpyHostToDevice));
    return 0;
}
int tvc_sac_get_adam_steps(tvc_sac* h, int32_t out[2]) {
    if (!h || !out) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    TVC_HIP_CHECK(hipDeviceSynchronize());
    flush_tick(h, nullptr);
    AdamClock c[2];
    TVC_HIP_CHECK(hipMemcpy(c, h->clk, sizeof(c), hipMemcpyDeviceToHost));
    out[0] = c[0].step; out[1] = c[1].step;
    return 0;
}
int tvc_sac_set_adam_steps(tvc_sac* h, const int32_t in[2]) {
    if (!h || !in) return tvc::set_error(TVC_EINVAL, "null argument");
    if (in[0] < 0 || in[1] < 0) return tvc::set_error(TVC_EINVAL, "negative step count");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    TVC_HIP_CHECK(hipDeviceSynchronize());
    h->tick_pending = false;
    return set_clocks(h, in);
}

// call counter of the train-mode acting passes (keys their dropout masks); checkpoints only, both calls synchronise
int tvc_sac_get_act_counter(tvc_sac* h, int32_t* out) {
    if (!h || !out) return tvc::set_error(TVC_EINVAL, "null argument");
    *out = 0;
    if (!h->act_ctr) return 0;
    TVC_HIP_CHECK(hipSetDevice(h->device));
    TVC_HIP_CHECK(hipMemcpy(out, h->act_ctr, sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}
int tvc_sac_set_act_counter(tvc_sac* h, int32_t value) {
    if (!h) return tvc::set_error(TVC_EINVAL, "null argument");
    if (!h->act_ctr) return 0;
    TVC_HIP_CHECK(hipSetDevice(h->device));
    TVC_HIP_CHECK(hipMemcpy(h->act_ctr, &value, sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

int tvc_sac_snapshot_policy(tvc_sac* h, void* stream) {
    if (!h) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    TVC_HIP_CHECK(hipMemcpyAsync(h->snap_p, h->P_actor(), h->n_actor * sizeof(float), hipMemcpyDeviceToDevice, st));
    TVC_HIP_CHECK(hipMemcpyAsync(h->snap_ov, h->ov, h->ov_floats * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (h->rows_ok) TVC_HIP_CHECK(hipMemcpyAsync(h->snap_pack, h->pack, h->pack_floats * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (h->train_rows_ok && h->train_stream_live)
        TVC_HIP_CHECK(hipMemcpyAsync(h->snap_tpack, h->tpack, h->tpack_floats * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (h->x3_live)
        TVC_HIP_CHECK(hipMemcpyAsync(h->snap_xpack, h->xpack, (size_t)h->x3_tiles * X3_TILE_BYTES, hipMemcpyDeviceToDevice, st));
    if (h->x3t_live)
        TVC_HIP_CHECK(hipMemcpyAsync(h->snap_xtpack, h->xtpack, (size_t)h->x3t_tiles * X3_TILE_BYTES, hipMemcpyDeviceToDevice, st));
    return 0;
}

int tvc_sac_sync_derived(tvc_sac* h, void* stream) {
    if (!h) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    refresh_folded(h, (hipStream_t)stream, nullptr);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_sac_actor_apply(tvc_sac* h, float grad_scale, void* stream) {
    if (!h) return tvc::set_error(TVC_EINVAL, "null argument");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    flush_tick(h, st);
    adam_apply(h, h->P_actor(), h->G_actor(), 0, h->n_actor, 1, grad_scale, true, st);  // + Polyak of the targets, same launch
    refresh_folded(h, st, h->clk + 1);                                                   // ... + the actor clock's advance
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_sac_update(tvc_sac* h, const float* s, const float* a, const float* r, const float* s2, const float* d,
                   const float* eps_next, const float* eps_new, float* losses, void* stream) {
    if (int e = tvc_sac_critic_grads(h, s, a, r, s2, d, eps_next, losses, stream)) return e;
    if (int e = tvc_sac_critic_apply(h, 1.0f, stream)) return e;
    if (int e = tvc_sac_actor_grads(h, s, eps_new, losses, stream)) return e;
    return tvc_sac_actor_apply(h, 1.0f, stream);
}

// Y[M,N] = act(X[M,K] W[N,K]^T + b): the fused Linear kernel on its own (numerics tests and kernel benchmarks).
// variant: 0 auto, 1 = 64x64 LDS-tiled, 3 = skinny split-K.
#ifdef TVC_GEMM_STAMPS
static unsigned long long* g_gemm_stamps = nullptr;
extern "C" void tvc_debug_set_gemm_stamps(unsigned long long* dev) { g_gemm_stamps = dev; }  // timing build only
#endif
int tvc_nn_linear_forward(const float* X, const float* W, const float* b, float* Y, int32_t M, int32_t N, int32_t K, int32_t act,
                          int32_t variant, void* stream) {
    if (!X || !W || !Y || M < 1 || N < 1 || K < 1) return tvc::set_error(TVC_EINVAL, "bad argument");
    GemmArgs g{};
    g.A = X; g.B = W; g.C = Y; g.M = M; g.N = N; g.K = K; g.K1 = K; g.lda = K; g.ldb = K; g.ldc = N; g.bias = b; g.act = act;
#ifdef TVC_GEMM_STAMPS
    g.stamps = g_gemm_stamps;
#endif
    g_force_variant = variant;
    launch_gemm(true, true, g, 1, (hipStream_t)stream);
    g_force_variant = 0;
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

// Y = LayerNorm(act(X W^T + b) + R): the fused acting kernel on its own (numerics tests, roofline measurement)
int tvc_nn_linear_ln_forward(const float* X, const float* W, const float* b, const float* R, const float* gamma, const float* beta,
                             float* Y, int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    if (!X || !W || !Y || !gamma || !beta || M < 1) return tvc::set_error(TVC_EINVAL, "bad argument");
    if ((N != 256 && N != 512) || K < GBK || (K % GBK) != 0) return tvc::set_error(TVC_EINVAL, "needs N in {256, 512}, K a multiple of 16");
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) & 15) return tvc::set_error(TVC_EINVAL, "X, W must be 16-byte aligned");
    RowLnArgs a{};
    a.A = X; a.B = W; a.C = Y; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N; a.bias = b; a.act = act; a.Radd = R;
    a.gamma = gamma; a.beta = beta;
    launch_rowln(a, (hipStream_t)stream);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

int tvc_sac_q_values(tvc_sac* h, const float* s, const float* a, int32_t n, int32_t use_target, float* q, void* stream) {
    if (!h || !s || !a || !q) return tvc::set_error(TVC_EINVAL, "null argument");
    if (n != h->cfg.batch_size) return tvc::set_error(TVC_EINVAL, "tvc_sac_q_values needs n == batch_size (%d)", h->cfg.batch_size);
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    In2 in2;
    const float* x = critic_input(h, s, a, in2, st);
    net_forward(h->critic, use_target ? h->P_tq() : h->P_q(), h->n_critic, x, 0, n, 2, h->cctx, false, nullptr, 0, st, nullptr,
                in2.X2 ? &in2 : nullptr);
    TVC_HIP_CHECK(hipMemcpyAsync(q, h->cctx.Y.back(), 2L * n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------ small MLP handle + curiosity / safety kernels (K9, K12)
struct tvc_mlp {
    NetDef net;
    Ctx ctx;
    const float* params;
    void* slab;
    float* xcat;
    float* tmp;
    int device, max_rows, in_dim, out_dim;
};

namespace {
// act_flags: low byte = hidden activation, TVC_MLP_LAYERNORM (0x100) = a LayerNorm behind every hidden activation
// (nn.Sequential numbering Linear, act, LayerNorm, Linear, ...: the goal policy of agent/...:366-374)
static NetDef build_mlp(const int32_t* dims, int n_layers, int act_flags) {
    NetDef n;
    n.in_dim = dims[0];
    const int act = act_flags & 0xff;
    const bool ln = (act_flags & TVC_MLP_LAYERNORM) != 0;
    const int per = ln ? 3 : 2;
    int x = 0;
    for (int l = 0; l < n_layers; ++l) {
        const bool last = l == n_layers - 1;
        const std::string name = std::to_string(per * l);
        if (last && dims[l + 1] <= 4 && (dims[l] % 4) == 0) x = n.add(OP_HEAD, name, dims[l], dims[l + 1], 0, x, -1, 0);
        else x = n.add(OP_LINEAR, name, dims[l], dims[l + 1], last ? ACT_NONE : act, x, -1, 0);
        if (ln && !last) x = n.add(OP_LN, std::to_string(per * l + 2), dims[l + 1], dims[l + 1], 0, x, -1, 0);
    }
    n.finish();
    return n;
}
static int mlp_dims_ok(const int32_t* dims, int n_layers) {
    if (!dims || n_layers < 1 || n_layers > 8) return tvc::set_error(TVC_EINVAL, "bad dims / n_layers");
    for (int l = 0; l <= n_layers; ++l)
        if (dims[l] < 1 || dims[l] > 4096) return tvc::set_error(TVC_EINVAL, "layer width out of range");
    return 0;
}
static int mlp_flags_ok(const int32_t* dims, int n_layers, int act_flags) {
    const int act = act_flags & 0xff;
    if (act < ACT_NONE || act > ACT_RELU || (act_flags & ~(0xff | TVC_MLP_LAYERNORM))) return tvc::set_error(TVC_EINVAL, "bad act / flags");
    if (act_flags & TVC_MLP_LAYERNORM)
        for (int l = 1; l < n_layers; ++l)
            if (dims[l] > 1024) return tvc::set_error(TVC_EINVAL, "LayerNorm widths up to 1024");
    return 0;
}

// softmax over the goal logits + one categorical draw per row from a uniform u in [0,1) (HierarchicalAgent.select_goal,
// agent/...:396-402), then the low-level policy's input [state | one-hot(goal)] (:404-413)
__global__ void goal_sample_kernel(const float* __restrict__ logits, const float* __restrict__ u, const float* __restrict__ state,
                                   int state_ld, int sd, int ng, int M, float* __restrict__ out, int* __restrict__ idx) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* lg = logits + (long)m * ng;
    float mx = lg[0];
    for (int j = 1; j < ng; ++j) mx = fmaxf(mx, lg[j]);
    float tot = 0.f;
    for (int j = 0; j < ng; ++j) tot += expf(lg[j] - mx);
    const float target = u[m] * tot;
    float cum = 0.f;
    int g = ng - 1;
    for (int j = 0; j < ng; ++j) {
        cum += expf(lg[j] - mx);
        if (target < cum) { g = j; break; }
    }
    float* o = out + (long)m * (sd + ng);
    for (int k = 0; k < sd; ++k) o[k] = state[(long)m * state_ld + k];
    for (int j = 0; j < ng; ++j) o[sd + j] = j == g ? 1.0f : 0.0f;
    if (idx) idx[m] = g;
}

// dense [n, k1 + k2] input from the first k1 columns of a (row stride lda) and the first k2 of b (row stride ldb)
__global__ void gather2_kernel(const float* __restrict__ a, int lda, int k1, const float* __restrict__ b, int ldb, int k2,
                               float* __restrict__ o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int K = k1 + k2;
    if (i >= n * K) return;
    const int m = i / K, k = i - m * K;
    o[i] = k < k1 ? a[(long)m * lda + k] : b[(long)m * ldb + (k - k1)];
}

// CuriosityModule.compute_intrinsic_reward (env/enhanced_rocket_tvc_env.py:257-269): rew += 0.01 * mean((pred - obs[:D])^2),
// skipped where skip[m] != 0 (first step of an episode, ref :496)
__global__ void curiosity_reward_kernel(const float* __restrict__ pred, const float* __restrict__ obs, int obs_ld,
                                        const unsigned char* __restrict__ skip, float* __restrict__ rew, int M, int D) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    if (skip && skip[m]) return;
    float e = 0.f;
    for (int j = 0; j < D; ++j) {
        const float d = pred[(long)m * D + j] - obs[(long)m * obs_ld + j];
        e += d * d;
    }
    rew[m] += 0.01f * e / (float)D;
}

// SafetyLayer.forward (agent/multi_algorithm_agent.py:304-351) + the clamp of get_action (:789)
__global__ void safety_select_kernel(const float* __restrict__ state, int sd, const float* __restrict__ proposed,
                                     const float* __restrict__ corr, float* __restrict__ out, int M, int A, float max_tilt,
                                     float max_w, float max_effort) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float* q = state + (long)m * sd;
    const float pitch = asinf(2.0f * (q[3] * q[1] - q[2] * q[0]));
    const float yaw = atan2f(2.0f * (q[3] * q[2] + q[0] * q[1]), 1.0f - 2.0f * (q[1] * q[1] + q[2] * q[2]));
    const float tilt = sqrtf(pitch * pitch + yaw * yaw);
    const float wn = sqrtf(q[4] * q[4] + q[5] * q[5] + q[6] * q[6]);
    float an = 0.f;
    for (int j = 0; j < A; ++j) an += proposed[(long)m * A + j] * proposed[(long)m * A + j];
    const bool viol = (tilt > max_tilt) || (wn > max_w) || (sqrtf(an) > max_effort);
    for (int j = 0; j < A; ++j) {
        const float v = viol ? corr[(long)m * A + j] : proposed[(long)m * A + j];
        out[(long)m * A + j] = fminf(fmaxf(v, -1.0f), 1.0f);
    }
}

// runs the MLP on [a[:, :k1] | b[:, :k2]]; returns the device pointer of the [n, out_dim] result (inside the handle)
static const float* mlp_run(tvc_mlp* h, const float* a, int lda, int k1, const float* b, int ldb, int n, hipStream_t st) {
    const int k2 = h->in_dim - k1;
    const float* in = a;
    In2 in2{b, 0, k1, lda, ldb};
    const bool thin = thin_ok(h->net.ops[0]);  // the first layer reads [a | b] in place
    if (!thin && (k2 > 0 || lda != h->in_dim)) {
        hipLaunchKernelGGL(gather2_kernel, dim3((n * h->in_dim + 255) / 256), dim3(256), 0, st, a, lda, k1, b, ldb, k2, h->xcat, n);
        in = h->xcat;
    }
    for (size_t i = 1; i < h->ctx.gY.size(); ++i) h->ctx.gY[i] = (long)n * h->net.buf_dim[i];
    net_forward(h->net, h->params, 0, in, 0, n, 1, h->ctx, false, nullptr, 0, st, nullptr, thin ? &in2 : nullptr);
    return h->ctx.Y.back();
}
}  // namespace

extern "C" {

int64_t tvc_mlp_param_count(const int32_t* dims, int32_t n_layers) {
    if (mlp_dims_ok(dims, n_layers)) return -1;
    return build_mlp(dims, n_layers, ACT_RELU).n_params;
}
int64_t tvc_mlp_layout(const int32_t* dims, int32_t n_layers, int32_t act_flags, int32_t layer, int64_t* w_off, int64_t* b_off,
                       int64_t* ln_w_off, int64_t* ln_b_off) {
    if (mlp_dims_ok(dims, n_layers) || mlp_flags_ok(dims, n_layers, act_flags)) return -1;
    NetDef n = build_mlp(dims, n_layers, act_flags);
    if (layer >= n_layers) { tvc::set_error(TVC_EINVAL, "layer out of range"); return -1; }
    if (layer >= 0) {
        const bool ln = (act_flags & TVC_MLP_LAYERNORM) != 0;
        const int oi = ln ? 2 * layer : layer;  // hidden layers own two ops (Linear, LayerNorm) when ln
        if (w_off) *w_off = n.ops[oi].w;
        if (b_off) *b_off = n.ops[oi].b;
        const bool has_ln = ln && layer < n_layers - 1;
        if (ln_w_off) *ln_w_off = has_ln ? n.ops[oi + 1].w : -1;
        if (ln_b_off) *ln_b_off = has_ln ? n.ops[oi + 1].b : -1;
    }
    return n.n_params;
}
int tvc_mlp_tensor_offset(const int32_t* dims, int32_t n_layers, int32_t layer, int64_t* w_off, int64_t* b_off) {
    if (int e = mlp_dims_ok(dims, n_layers)) return e;
    if (layer < 0 || layer >= n_layers) return tvc::set_error(TVC_EINVAL, "layer out of range");
    NetDef n = build_mlp(dims, n_layers, ACT_RELU);
    if (w_off) *w_off = n.ops[layer].w;
    if (b_off) *b_off = n.ops[layer].b;
    return 0;
}
int tvc_mlp_create(const int32_t* dims, int32_t n_layers, int32_t act, int32_t max_rows, int32_t device, const float* params_dev,
                   tvc_mlp** out) {
    if (!out) return tvc::set_error(TVC_EINVAL, "out is NULL");
    *out = nullptr;
    if (int e = mlp_dims_ok(dims, n_layers)) return e;
    if (!params_dev || max_rows < 1) return tvc::set_error(TVC_EINVAL, "bad argument");
    if (int e = mlp_flags_ok(dims, n_layers, act)) return e;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return tvc::set_error(TVC_ENODEV, "no HIP device visible: libtvc_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return tvc::set_error(TVC_EINVAL, "device out of range");
    TVC_HIP_CHECK(hipSetDevice(device));
    tvc_mlp* h = new (std::nothrow) tvc_mlp();
    if (!h) return tvc::set_error(TVC_ENOMEM, "host allocation failed");
    h->net = build_mlp(dims, n_layers, act);
    h->params = params_dev; h->device = device; h->max_rows = max_rows; h->in_dim = dims[0]; h->out_dim = dims[n_layers];
    int maxd = dims[0];
    for (int l = 1; l <= n_layers; ++l) maxd = std::max(maxd, (int)dims[l]);
    const long bytes = 10L * (((long)max_rows * maxd * 4 + 255) & ~255L) + 4096;
    hipError_t he = hipMalloc(&h->slab, bytes);
    if (he != hipSuccess) {
        delete h;
        return tvc::set_error(TVC_ENOMEM, "hipMalloc(%ld) failed: %s", bytes, hipGetErrorString(he));
    }
    char* p = (char*)h->slab;
    h->xcat = (float*)carve(p, (long)max_rows * maxd * 4);
    h->tmp = (float*)carve(p, (long)max_rows * maxd * 4);
    if (ctx_alloc_infer(h->ctx, h->net, max_rows, p) != 0) {
        (void)hipFree(h->slab);
        delete h;
        return tvc::set_error(TVC_EINVAL, "slot allocation failed");
    }
    *out = h;
    return 0;
}
void tvc_mlp_destroy(tvc_mlp* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->slab);
    delete h;
}
int tvc_mlp_forward(tvc_mlp* h, const float* x, int32_t x_ld, int32_t k1, const float* x2, int32_t x2_ld, int32_t n, float* out,
                    void* stream) {
    if (!h || !x || !out) return tvc::set_error(TVC_EINVAL, "null argument");
    if (n < 1 || n > h->max_rows || k1 < 1 || k1 > h->in_dim || (k1 < h->in_dim && !x2)) return tvc::set_error(TVC_EINVAL, "bad n / k1");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const float* y = mlp_run(h, x, x_ld, k1, x2, x2_ld, n, st);
    TVC_HIP_CHECK(hipMemcpyAsync(out, y, (long)n * h->out_dim * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}
// rew[m] += 0.01 * mean((f([prev_obs[m, :D] | act[m]]) - obs[m, :D])^2) unless skip[m]   (env/...:496-502)
int tvc_curiosity_add(tvc_mlp* h, const float* prev_obs, int32_t obs_ld, const float* act, int32_t act_dim, const float* obs,
                      const uint8_t* skip, float* rew, int32_t n, void* stream) {
    if (!h || !prev_obs || !act || !obs || !rew) return tvc::set_error(TVC_EINVAL, "null argument");
    const int D = h->out_dim;
    if (n < 1 || n > h->max_rows || D + act_dim != h->in_dim || D > obs_ld) return tvc::set_error(TVC_EINVAL, "shape mismatch");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const float* pred = mlp_run(h, prev_obs, obs_ld, D, act, act_dim, n, st);
    hipLaunchKernelGGL(curiosity_reward_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pred, obs, obs_ld, skip, rew, n, D);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}
// out = clamp(violates(state, proposed) ? safety_net([state | proposed]) : proposed, -1, 1)   (agent/...:304-351, :789)
int tvc_safety_apply(tvc_mlp* h, const float* state, int32_t state_dim, const float* proposed, float* out, int32_t n, float max_tilt,
                     float max_w, float max_effort, void* stream) {
    if (!h || !state || !proposed || !out) return tvc::set_error(TVC_EINVAL, "null argument");
    const int A = h->out_dim;
    if (n < 1 || n > h->max_rows || state_dim + A != h->in_dim || state_dim < 7) return tvc::set_error(TVC_EINVAL, "shape mismatch");
    TVC_HIP_CHECK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const float* corr = mlp_run(h, state, state_dim, state_dim, proposed, A, n, st);
    hipLaunchKernelGGL(safety_select_kernel, dim3((n + 255) / 256), dim3(256), 0, st, state, state_dim, proposed, corr, out, n, A,
                       max_tilt, max_w, max_effort);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}
int tvc_goal_sample(const float* logits, const float* u, const float* state, int32_t state_ld, int32_t state_dim, int32_t n_goals,
                    int32_t n, float* state_goal_out, int32_t* goal_idx_out, void* stream) {
    if (!logits || !u || !state || !state_goal_out) return tvc::set_error(TVC_EINVAL, "null argument");
    if (n < 1 || n_goals < 1 || n_goals > 64 || state_dim < 1 || state_ld < state_dim) return tvc::set_error(TVC_EINVAL, "bad shape");
    hipLaunchKernelGGL(goal_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, u, state, state_ld,
                       state_dim, n_goals, n, state_goal_out, goal_idx_out);
    TVC_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
