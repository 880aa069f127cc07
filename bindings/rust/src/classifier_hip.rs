//! `predict_batch_with_context` of the reference (src/classifier.rs:826-867) on top of the C ABI.
//! Source only -- see ../README.md.
use crate::ffi::*;
use crate::{BatchInferenceContext, Error, InferenceOptions, Prediction, PredictionResult, Result};
use std::sync::atomic::Ordering;

impl crate::Classifier {
    pub fn predict_batch_with_context(
        &self,
        context: &mut BatchInferenceContext,
        segments: &[&[f32]],
        options: &InferenceOptions,
    ) -> Result<Vec<PredictionResult>> {
        if segments.is_empty() {
            return Ok(Vec::new());
        }
        let n = segments.len();
        // batch_context.rs:191-206: batch limit first, then per-segment sizes
        if n > context.max_batch_size() {
            return Err(Error::Inference(format!("batch size {} exceeds context max {}", n, context.max_batch_size())));
        }
        for (i, s) in segments.iter().enumerate() {
            if s.len() != context.sample_count() {
                return Err(Error::BatchInputSize { index: i, expected: context.sample_count(), got: s.len() });
            }
        }
        let cfg = &self.inner.config;
        let (nsp, emb_dim) = (cfg.num_species, cfg.embedding_dim.unwrap_or(0));
        let ptrs: Vec<*const f32> = segments.iter().map(|s| s.as_ptr()).collect();
        let mut logits = vec![0f32; n * nsp];
        let mut emb = vec![0f32; n * emb_dim];
        let cancel = options.cancellation_token.as_ref().map_or(std::ptr::null(), |t| t.cancelled.as_ptr() as *const i32);
        let timeout_ns = options.timeout.map_or(0, |d| d.as_nanos().max(1) as u64);
        let st = unsafe {
            bn_infer(context.ctx, ptrs.as_ptr(), n, logits.as_mut_ptr(),
                     if emb_dim > 0 { emb.as_mut_ptr() } else { std::ptr::null_mut() }, cancel, timeout_ns)
        };
        match st {
            BN_OK => {}
            BN_ERR_TIMEOUT => return Err(Error::Timeout { duration: options.timeout.unwrap_or_default() }),
            BN_ERR_CANCELLED => return Err(Error::Cancelled),
            _ => return Err(Error::Inference(last_error())),
        }
        let k = self.inner.top_k.min(nsp);
        let (mut idx, mut conf, mut cnt) = (vec![0u32; n * k.max(1)], vec![0f32; n * k.max(1)], vec![0u32; n]);
        let (has_min, min) = self.inner.min_confidence.map_or((0, 0.0), |m| (1, m));
        if unsafe { bn_topk(context.ctx, n, self.inner.top_k, has_min, min, k.max(1), idx.as_mut_ptr(), conf.as_mut_ptr(), cnt.as_mut_ptr()) } != BN_OK {
            return Err(Error::Inference(last_error()));
        }
        Ok(self.assemble_results(n, k.max(1), &logits, &emb, &idx, &conf, &cnt))
    }

    /// `process_batch_outputs_from_flat` of the reference (src/classifier.rs:872-911): flat logits / embeddings and the
    /// device-side top-K rows (index, confidence, count per segment; row stride `k`) -> one `PredictionResult` per segment.
    fn assemble_results(&self, n: usize, k: usize, logits: &[f32], emb: &[f32], idx: &[u32], conf: &[f32], cnt: &[u32]) -> Vec<PredictionResult> {
        let cfg = &self.inner.config;
        let nsp = cfg.num_species;
        (0..n)
            .map(|i| PredictionResult {
                model_type: cfg.model_type,
                predictions: (0..cnt[i] as usize)
                    .map(|j| {
                        let index = idx[i * k + j] as usize;
                        Prediction {
                            species: self.inner.labels.get(index).cloned().unwrap_or_else(|| format!("unknown_{index}")),
                            confidence: conf[i * k + j],
                            index,
                        }
                    })
                    .collect(),
                embeddings: cfg.embedding_dim.map(|d| emb[i * d..(i + 1) * d].to_vec()),
                raw_scores: logits[i * nsp..(i + 1) * nsp].to_vec(),
            })
            .collect()
    }
}

/// Pipelined variant for a stream of batches (e.g. the CLI's batch loop, birdnet-analyze.rs:556-600): two batches in
/// flight on ONE context -- while batch k runs on the GPU, batch k+1 is being copied into pinned staging and uploaded.
/// Top-K runs on the device as part of the submitted work; `collect` returns the same `PredictionResult`s as
/// `predict_batch_with_context`.  More overlap = more contexts (up to four per device), each driven like this.
impl crate::Classifier {
    pub fn submit_batch(&self, context: &mut BatchInferenceContext, segments: &[&[f32]]) -> Result<u64> {
        if segments.len() > context.max_batch_size() {
            return Err(Error::Inference(format!("batch size {} exceeds context max {}", segments.len(), context.max_batch_size())));
        }
        for (i, s) in segments.iter().enumerate() {
            if s.len() != context.sample_count() {
                return Err(Error::BatchInputSize { index: i, expected: context.sample_count(), got: s.len() });
            }
        }
        let ptrs: Vec<*const f32> = segments.iter().map(|s| s.as_ptr()).collect();
        let (has_min, min) = self.inner.min_confidence.map_or((0, 0.0), |m| (1, m));
        let mut ticket = 0u64;
        match unsafe { bn_infer_submit(context.ctx, ptrs.as_ptr(), segments.len(), self.inner.top_k, has_min, min, &mut ticket) } {
            BN_OK => Ok(ticket), // the slices are no longer referenced: they were copied into pinned staging
            _ => Err(Error::Inference(last_error())),
        }
    }

    /// Results of a submitted batch of `n` segments, as `predict_batch_with_context` returns them (classifier.rs:826-867).
    pub fn collect_batch(&self, context: &mut BatchInferenceContext, ticket: u64, n: usize, options: &InferenceOptions)
        -> Result<Vec<PredictionResult>> {
        let cfg = &self.inner.config;
        let (nsp, emb_dim, k) = (cfg.num_species, cfg.embedding_dim.unwrap_or(0), self.inner.top_k.min(cfg.num_species).max(1));
        let (mut logits, mut emb) = (vec![0f32; n * nsp], vec![0f32; n * emb_dim]);
        let (mut idx, mut conf, mut cnt) = (vec![0u32; n * k], vec![0f32; n * k], vec![0u32; n]);
        let cancel = options.cancellation_token.as_ref().map_or(std::ptr::null(), |t| t.cancelled.as_ptr() as *const i32);
        let timeout_ns = options.timeout.map_or(0, |d| d.as_nanos().max(1) as u64);
        match unsafe {
            bn_infer_collect(context.ctx, ticket, logits.as_mut_ptr(), if emb_dim > 0 { emb.as_mut_ptr() } else { std::ptr::null_mut() },
                             k, idx.as_mut_ptr(), conf.as_mut_ptr(), cnt.as_mut_ptr(), cancel, timeout_ns)
        } {
            BN_OK => Ok(self.assemble_results(n, k, &logits, &emb, &idx, &conf, &cnt)),
            BN_ERR_TIMEOUT => Err(Error::Timeout { duration: options.timeout.unwrap_or_default() }),
            BN_ERR_CANCELLED => Err(Error::Cancelled),
            _ => Err(Error::Inference(last_error())),
        }
    }
}

fn last_error() -> String {
    let mut buf = vec![0u8; 1024];
    unsafe { bn_last_error(buf.as_mut_ptr() as *mut _, buf.len()) };
    String::from_utf8_lossy(&buf).trim_end_matches('\0').to_string()
}

// ---- the rest of the Classifier surface on the same ABI (source only, mirrors csrc/host_classifier.cpp) ----------------

/// `ClassifierBuilder::build` (reference src/classifier.rs:334-383): same validation order and error variants; the ORT
/// session becomes a `bn_model`, the model type / shapes come from `bn_model_get_config` (same rules as detection.rs).
impl crate::ClassifierBuilder {
    pub fn build_hip(self, device: i32) -> Result<crate::Classifier> {
        crate::ffi::assert_abi();
        let model_path = self.model_path.ok_or(Error::ModelPathRequired)?;
        if self.labels.is_none() && self.labels_path.is_none() {
            return Err(Error::LabelsRequired);
        }
        let cpath = std::ffi::CString::new(model_path.as_str()).map_err(|e| Error::ModelLoad(e.to_string()))?;
        let mut model: *mut bn_model = std::ptr::null_mut();
        let override_ = self.model_type.map_or(-1, |t| t as i32);
        match unsafe { bn_model_load(cpath.as_ptr(), device, override_, &mut model) } {
            BN_OK => {}
            BN_ERR_MODEL_DETECTION => return Err(Error::ModelDetection { reason: last_error() }),
            _ => return Err(Error::ModelLoad(last_error())),
        }
        let mut cfg = bn_model_config::default();
        unsafe { bn_model_get_config(model, &mut cfg) };
        let config = crate::ModelConfig::from_native(&cfg); // model_type, sample_rate, segment_duration, sample_count, num_species, embedding_dim
        let labels = match self.labels {
            Some(v) => v,
            None => crate::labels::load_labels_from_file(self.labels_path.as_ref().unwrap(), config.model_type)?,
        };
        if labels.len() != config.num_species {
            unsafe { bn_model_free(model) };
            return Err(Error::LabelCount { expected: config.num_species, got: labels.len() });
        }
        // the default context the reference keeps behind its `Mutex<Session>` (classifier.rs:435)
        let mut ctx: *mut bn_ctx = std::ptr::null_mut();
        if unsafe { bn_ctx_create(model, 1, 0, &mut ctx) } != BN_OK {
            unsafe { bn_model_free(model) };
            return Err(Error::ModelLoad(last_error()));
        }
        Ok(crate::Classifier::from_native(model, ctx, config, labels, self.top_k, self.min_confidence))
    }
}

impl crate::Classifier {
    /// `predict` (classifier.rs:610-643): one segment; the size check stays in Rust, the default context grows on demand.
    pub fn predict_with_options(&self, segment: &[f32], options: &InferenceOptions) -> Result<PredictionResult> {
        if segment.len() != self.inner.config.sample_count {
            return Err(Error::InputSize { expected: self.inner.config.sample_count, got: segment.len() });
        }
        let mut guard = self.inner.default_context.lock().map_err(|_| Error::Inference("context lock poisoned".into()))?;
        guard.ensure_capacity(self, 1)?; // re-creates the bn_ctx when max_batch is too small
        let mut out = self.predict_batch_with_context(&mut guard, &[segment], options)?;
        Ok(out.remove(0))
    }

    /// `predict_batch` (classifier.rs:676-727): empty input -> empty output before anything else, per-segment size errors
    /// carry the index (`BatchInputSize`), then one pass through the default context.
    pub fn predict_batch_with_options(&self, segments: &[&[f32]], options: &InferenceOptions) -> Result<Vec<PredictionResult>> {
        if segments.is_empty() {
            return Ok(Vec::new());
        }
        let mut guard = self.inner.default_context.lock().map_err(|_| Error::Inference("context lock poisoned".into()))?;
        guard.ensure_capacity(self, segments.len())?;
        self.predict_batch_with_context(&mut guard, segments, options)
    }

    /// `create_batch_context` (classifier.rs:777-792, batch_context.rs:70-133): the reference refuses Perch v2 here and so
    /// does this shim; `create_native_batch_context` lifts the refusal (the native context runs Perch).
    pub fn create_batch_context(&self, max_batch_size: usize) -> Result<BatchInferenceContext> {
        if self.inner.config.model_type == crate::ModelType::PerchV2 {
            // the reference's text, src/batch_context.rs:110-113
            return Err(Error::Inference(
                "BatchInferenceContext does not yet support PerchV2 models. Use predict_batch() instead.".into(),
            ));
        }
        self.create_native_batch_context(max_batch_size)
    }

    pub fn create_native_batch_context(&self, max_batch_size: usize) -> Result<BatchInferenceContext> {
        let mut ctx: *mut bn_ctx = std::ptr::null_mut();
        match unsafe { bn_ctx_create(self.inner.model, max_batch_size, 0, &mut ctx) } {
            BN_OK => Ok(BatchInferenceContext::from_native(ctx, max_batch_size, self.inner.config.sample_count)),
            _ => Err(Error::Inference(last_error())),
        }
    }
}

impl Drop for BatchInferenceContext {
    fn drop(&mut self) {
        unsafe { bn_ctx_destroy(self.ctx) } // the model is reference-counted on the native side: order does not matter
    }
}

// ---- rows f3 / f4 / a10 of SURVEY.md section 8 from Rust (round 5) ------------------------------------------------------

impl crate::Classifier {
    /// The native context with every graph output kept (`BN_CTX_ALL_OUTPUTS`): Perch v2's spectrogram `[500, 128]` and spatial
    /// embedding `[16, 4, 1536]` (detection.rs:58-71 names them outputs 2 and 1) stay readable after a run.
    pub fn create_native_batch_context_all_outputs(&self, max_batch_size: usize) -> Result<BatchInferenceContext> {
        let mut ctx: *mut bn_ctx = std::ptr::null_mut();
        match unsafe { bn_ctx_create(self.inner.model, max_batch_size, BN_CTX_ALL_OUTPUTS, &mut ctx) } {
            BN_OK => Ok(BatchInferenceContext::from_native(ctx, max_batch_size, self.inner.config.sample_count)),
            _ => Err(Error::Inference(last_error())),
        }
    }

    /// What `session.inputs()` / `session.outputs()` report (classifier.rs:387-420): input shape and every output's shape, a
    /// dynamic dimension as -1 -- the facts `detect_model_type` (detection.rs:15-145) decides on.
    pub fn io_info(&self) -> Result<(Vec<i64>, Vec<Vec<i64>>)> {
        let mut info = bn_io_info::default();
        if unsafe { bn_model_io_info(self.inner.model, &mut info) } != BN_OK {
            return Err(Error::Inference(last_error()));
        }
        let input = info.input_shape[..info.input_rank as usize].to_vec();
        let outputs = (0..info.n_outputs as usize).map(|o| info.output_shape[o][..info.output_rank[o] as usize].to_vec()).collect();
        Ok((input, outputs))
    }
}

impl BatchInferenceContext {
    /// Graph output `index` of the last run, `[batch, row]` row-major (the native counterpart of
    /// `extract_tensor_data`, classifier.rs:1062-1077, for outputs the reference discards).
    pub fn read_output(&mut self, index: i32, batch: usize) -> Result<Vec<f32>> {
        let (mut d, mut row) = (std::ptr::null::<f32>(), 0usize);
        if unsafe { bn_ctx_output_device(self.ctx, index, &mut d, &mut row) } != BN_OK {
            return Err(Error::Inference(last_error()));
        }
        let mut host = vec![0f32; batch * row];
        match unsafe { bn_ctx_read_output(self.ctx, index, batch, host.as_mut_ptr()) } {
            BN_OK => Ok(host),
            _ => Err(Error::Inference(last_error())),
        }
    }

    /// Graph captures / replays / fallbacks of this context (`capture_fallbacks` must stay 0).
    pub fn stats(&self) -> bn_ctx_stats {
        let mut st = bn_ctx_stats::default();
        unsafe { bn_ctx_get_stats(self.ctx, &mut st, std::mem::size_of::<bn_ctx_stats>()) };
        st
    }
}

/// A recording resident on the device in its storage format (int16 or f32), cut into windows by `chunk_audio`'s rules
/// (birdnet-analyze.rs:707-743) inside the first kernel.  `new_resampled` converts a recording of another rate to the
/// model's rate on the device -- the reference's CLI rejects such files (birdnet-analyze.rs:447-455).
pub struct Recording {
    pub(crate) rec: *mut bn_recording,
}

impl Recording {
    pub fn from_i16(device: i32, pcm: &[i16]) -> Result<Self> {
        let mut rec: *mut bn_recording = std::ptr::null_mut();
        match unsafe { bn_recording_create(device, pcm.as_ptr() as *const _, pcm.len(), BN_PCM_I16, &mut rec) } {
            BN_OK => Ok(Self { rec }),
            _ => Err(Error::Inference(last_error())),
        }
    }

    pub fn new_resampled(device: i32, pcm: &[f32], src_rate: u32, dst_rate: u32) -> Result<Self> {
        let mut rec: *mut bn_recording = std::ptr::null_mut();
        match unsafe { bn_recording_create_resampled(device, pcm.as_ptr() as *const _, pcm.len(), BN_PCM_F32, src_rate, dst_rate, 0, &mut rec) } {
            BN_OK => Ok(Self { rec }),
            _ => Err(Error::Inference(last_error())),
        }
    }

    /// Blocks until an asynchronous upload has landed (a no-op for the synchronous constructors).
    pub fn wait(&self) -> Result<()> {
        match unsafe { bn_recording_wait(self.rec) } {
            BN_OK => Ok(()),
            _ => Err(Error::Inference(last_error())),
        }
    }

    pub fn samples(&self) -> usize {
        unsafe { bn_recording_samples(self.rec) }
    }

    /// Windows `chunk_audio` would cut at this step (`step = sample_count - overlap * sample_rate`).
    pub fn n_windows(&self, step_samples: usize) -> usize {
        unsafe { bn_chunk_count(self.samples(), step_samples) }
    }
}

impl Drop for Recording {
    fn drop(&mut self) {
        unsafe { bn_recording_free(self.rec) }
    }
}

/// One node, several GPUs (BASELINE configs[4]): windows sharded by contiguous range, logits gathered over RCCL.
pub struct Group {
    pub(crate) g: *mut bn_group,
}

impl Group {
    pub fn size(&self) -> i32 {
        unsafe { bn_group_size(self.g) }
    }

    pub fn uses_rccl(&self) -> bool {
        unsafe { bn_group_uses_rccl(self.g) != 0 }
    }

    /// Counters of every context of every rank, summed.
    pub fn stats(&self) -> Result<bn_ctx_stats> {
        let mut st = bn_ctx_stats::default();
        match unsafe { bn_group_get_stats(self.g, &mut st, std::mem::size_of::<bn_ctx_stats>()) } {
            BN_OK => Ok(st),
            _ => Err(Error::Inference(group_last_error())),
        }
    }
}

impl Drop for Group {
    fn drop(&mut self) {
        unsafe { bn_group_destroy(self.g) }
    }
}

fn group_last_error() -> String {
    let mut buf = vec![0u8; 1024];
    unsafe { bn_group_last_error(buf.as_mut_ptr() as *mut _, buf.len()) };
    String::from_utf8_lossy(&buf).trim_end_matches('\0').to_string()
}

/// How launches size their grids where one launch's latency trades against the work per block (`bn_set_sharing_mode`): `Alone` (default)
/// for the lowest latency of one batch, `Shared` for a caller that keeps several `BatchInferenceContext`s busy (identical results, bit
/// for bit; BirdNET v2.4 with four contexts: +7 % segments/s), `Auto` = `Shared` while more than one context lives on the device.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum SharingMode {
    Auto,
    Alone,
    Shared,
}

pub fn set_sharing_mode(mode: SharingMode) {
    let m = match mode {
        SharingMode::Auto => BN_SHARING_AUTO,
        SharingMode::Alone => BN_SHARING_ALONE,
        SharingMode::Shared => BN_SHARING_SHARED,
    };
    unsafe { bn_set_sharing_mode(m) }
}
