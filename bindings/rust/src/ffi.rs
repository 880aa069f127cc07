//! Raw bindings of include/birdnet_hip.h (BN_ABI_VERSION 2: checked by `assert_abi`).  Source only -- see ../README.md.
#![allow(non_camel_case_types)]
use std::os::raw::c_char;

#[repr(C)]
pub struct bn_model { _p: [u8; 0] }
#[repr(C)]
pub struct bn_recording { _p: [u8; 0] }
#[repr(C)]
pub struct bn_group { _p: [u8; 0] }
#[repr(C)]
pub struct bn_ctx { _p: [u8; 0] }

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct bn_model_config {
    pub model_type: i32,
    pub sample_rate: u32,
    pub segment_duration: f32,
    pub sample_count: u64,
    pub num_species: u64,
    pub has_embedding: i32,
    pub embedding_dim: u64,
    pub logits_output: i32,
    pub embedding_output: i32,
}

/// `bn_ctx_get_stats`: captures / replays / capture fallbacks (must be 0) of one context.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct bn_ctx_stats {
    pub captures: u64,
    pub instantiates: u64,
    pub replays: u64,
    pub eager_runs: u64,
    pub capture_fallbacks: u64,
    pub evictions: u64,
    pub cached_graphs: u64,
    pub last_fallback: [c_char; 192],
    pub input_copies: u64,
}

pub const BN_ABI_VERSION: i32 = 2;
/// Call once before anything else: a library built from another header revision is refused instead of misread.
pub fn assert_abi() {
    let got = unsafe { bn_abi_version() };
    assert_eq!(got, BN_ABI_VERSION, "libbirdnet_hip speaks ABI {got}, this crate was written for ABI {BN_ABI_VERSION}");
}

pub const BN_OK: i32 = 0;
pub const BN_ERR_TIMEOUT: i32 = 3;
pub const BN_ERR_CANCELLED: i32 = 4;
pub const BN_ERR_MODEL_DETECTION: i32 = 8;
pub const BN_ERR_NO_DEVICE: i32 = 9;

#[link(name = "birdnet_hip")]
extern "C" {
    pub fn bn_abi_version() -> i32;
    pub fn bn_model_load(path: *const c_char, device: i32, model_type_override: i32, out: *mut *mut bn_model) -> i32;
    pub fn bn_model_free(m: *mut bn_model);
    pub fn bn_model_get_config(m: *const bn_model, out: *mut bn_model_config) -> i32;
    pub fn bn_ctx_create(m: *mut bn_model, max_batch: usize, flags: u32, out: *mut *mut bn_ctx) -> i32;
    pub fn bn_ctx_destroy(c: *mut bn_ctx);
    pub fn bn_ctx_get_stats(c: *const bn_ctx, out: *mut bn_ctx_stats, struct_size: usize) -> i32;
    pub fn bn_ctx_input_device(c: *const bn_ctx, d_ptr: *mut *mut f32, capacity_floats: *mut usize) -> i32;
    pub fn bn_infer(c: *mut bn_ctx, segs: *const *const f32, batch: usize, logits_out: *mut f32, emb_out: *mut f32,
                    cancel: *const i32, timeout_ns: u64) -> i32;
    pub fn bn_topk(c: *mut bn_ctx, batch: usize, top_k: usize, has_min: i32, min_conf: f32, k_stride: usize,
                   idx_out: *mut u32, conf_out: *mut f32, count_out: *mut u32) -> i32;
    pub fn bn_last_error(buf: *mut c_char, cap: usize) -> usize;
    // asynchronous host-slice path: two batches in flight per context (staging + upload of batch k+1 overlap batch k)
    pub fn bn_infer_submit(c: *mut bn_ctx, segs: *const *const f32, batch: usize, top_k: usize, has_min: i32, min_conf: f32,
                           ticket: *mut u64) -> i32;
    pub fn bn_infer_collect(c: *mut bn_ctx, ticket: u64, logits_out: *mut f32, emb_out: *mut f32, k_stride: usize,
                            idx_out: *mut u32, conf_out: *mut f32, count_out: *mut u32, cancel: *const i32, timeout_ns: u64) -> i32;
    // one node, several GPUs: windows sharded by contiguous range, one RCCL all-gather of logits / top-K rows
    pub fn bn_group_create(models: *const *mut bn_model, devices: *const i32, n: i32, max_batch: usize,
                           contexts_per_device: i32, out: *mut *mut bn_group) -> i32;
    pub fn bn_group_destroy(g: *mut bn_group);
    pub fn bn_group_size(g: *const bn_group) -> i32;
    pub fn bn_group_uses_rccl(g: *const bn_group) -> i32;
    pub fn bn_shard_range(n_windows: usize, rank: i32, world: i32, lo: *mut usize, hi: *mut usize);
    pub fn bn_group_analyze_recording(g: *mut bn_group, pcm: *const core::ffi::c_void, n_samples: usize, format: i32,
                                      step_samples: usize, top_k: usize, has_min: i32, min_conf: f32, logits_out: *mut f32,
                                      k_stride: usize, idx_out: *mut u32, conf_out: *mut f32, count_out: *mut u32,
                                      n_windows_out: *mut usize) -> i32;
    pub fn bn_group_last_error(buf: *mut c_char, cap: usize) -> usize;
    // recording-level ingest (optional; birdnet-analyze.rs read_wav + chunk_audio on the device)
    pub fn bn_recording_create(device: i32, pcm: *const core::ffi::c_void, n_samples: usize, format: i32, out: *mut *mut bn_recording) -> i32;
    /// returns at once; `pcm` must outlive `bn_recording_wait` / `bn_recording_free` (the windows' calls wait for the samples they read)
    pub fn bn_recording_create_async(device: i32, pcm: *const core::ffi::c_void, n_samples: usize, format: i32, out: *mut *mut bn_recording) -> i32;
    pub fn bn_recording_wait(r: *const bn_recording) -> i32;
    pub fn bn_recording_free(r: *mut bn_recording);
    pub fn bn_chunk_count(n_samples: usize, step_samples: usize) -> usize;
    pub fn bn_infer_windows(c: *mut bn_ctx, r: *const bn_recording, step_samples: usize, first_window: usize, count: usize,
                            logits_out: *mut f32, emb_out: *mut f32, cancel: *const i32, timeout_ns: u64) -> i32;
}
