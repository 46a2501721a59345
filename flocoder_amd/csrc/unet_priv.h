// The velocity U-Net object behind the fc_unet_* entry points (unet.hip: parameters, forward plan, integrator;
// unet_backward.hip: backward plan of the training step).
#pragma once
#include <map>
#include <tuple>

#include "plan.h"

struct fc_unet : fc::ParamStore {
    fc_unet_config cfg{};
    int device = 0;
    int td = 0, heads = 4;
    std::vector<int> chans;  // [dim, dim*m0, dim*m1, ...]
    int S = 0;                                    // total scale/shift width
    std::unordered_map<std::string, int> ss_off;  // resblock prefix -> column offset
    float* freqs = nullptr;

    // plans: the batch can run as `nchains` independent row ranges on concurrent streams (no cross-sample op exists in the
    // network; FLOCODER_AMD_CHAINS=2).  Off by default: half-batch launches lose more than the overlap wins on one GPU.
    int maxB = 0, H = 0, W = 0, nchains = 1;
    fc::Plan plan[2];
    std::vector<void*> int_allocs;           // integrator state
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    // integrator state (library-owned so captured graphs never see caller pointers)
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    int* step = nullptr;
    float *ts_dev = nullptr, *sc = nullptr, *tvec = nullptr;
    int ts_cap = 0;
    float *y = nullptr, *xs = nullptr, *k1 = nullptr, *k2 = nullptr, *k3 = nullptr, *v2 = nullptr, *mask_own = nullptr;
    int64_t* ids_own = nullptr;
    float* pre = nullptr;                    // conditioning of every evaluation of the running integration (CondFetch): tv | t_emb | h | c1 | ss
    size_t pre_cap = 0;                      // floats
    float* pre_ss = nullptr;                 // the [evaluation][row][S] part of `pre`
    fc::TembArgs temb_proto;                 // weights of the conditioning chain as the plan's own launches use them
    std::map<std::tuple<int, int, int, int, uint32_t, uint32_t, uint32_t, int>, hipGraphExec_t> graphs;

    // Fused Block tails whose workgroups wait for each other (conv_dev.h) need the device to themselves.  `shared` = the caller said the
    // device is shared with other streams / processes (fc_unet_set_shared): plans are then built without such launches.  A wait that
    // times out anyway poisons its sample group with NaN and sets `dev_err`; `host_err` (pinned) receives a copy behind every forward /
    // integration, and every entry point refuses to go on once it is set (sticky until the plan is rebuilt).
    bool shared = false;
    int* dev_err = nullptr;
    volatile int* host_err = nullptr;
    bool tail_failed = false;
    hipEvent_t ev_meet = nullptr;            // end of this handle's last plan with meeting launches (process-wide guard, unet.hip)

    bool keep_all = false;   // plans keep every intermediate (q/k/v, attention output) for the backward: set by fc_unet_train_reserve

    // training (unet_backward.hip): backward launch plan over the forward arena, data-gradient weight operands
    fc::Plan bwd;
    struct DgradPack { int64_t src; float* dst; int O, I, KS, ci0, nci; };
    std::vector<DgradPack> dgrad_packs;
    fc::PackTable dgrad_table;                // all of them as one launch
    uint64_t param_version = 0, dgrad_version = ~0ull;
    int64_t class_lo = 0, class_hi = 0;       // [lo, hi) of class_cond_mlp.* in the flat table
    // gradient buckets (fc_unet_backward_parts): backward plan entries [0, bwd_split_op) leave [grad_split, end) of the flat gradient vector
    // complete (final_*, mid_*, ups.*); the rest of the plan completes [0, grad_split).  bwd_split_op < 0: one bucket
    int bwd_split_op = -1;
    int64_t grad_split = 0;
    bool want_buckets = false;                // fc_unet_set_grad_buckets: build the backward plan in its two-bucket form (data-parallel trainers)

    // What the activation arena currently holds.  fc_unet_backward_ex reads the activations the LAST forward left there, so every
    // entry point that writes the arena moves `arena_serial`; `arena_train_rows` > 0 only after a forward on the keep-everything
    // (training) plan with that many rows.  A backward that does not follow such a forward fails with FC_E_STATE instead of
    // producing gradients from someone else's activations.
    uint64_t arena_serial = 0;
    int arena_train_rows = 0;
    void arena_touched(int train_rows) { ++arena_serial; arena_train_rows = train_rows; }

};
