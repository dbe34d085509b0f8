//! `stringwars_amd` -- safe, `szs`-shaped wrappers over the C ABI of `libstringwars_amd.so`
//! (`include/stringwars_amd.h`), the MI355X-native batched Levenshtein / Needleman-Wunsch / Smith-Waterman backend.
//!
//! The type and method names follow what `similarities/bench.rs` uses from `stringzilla::szs`
//! (`DeviceScope`, `LevenshteinDistances[Utf8]`, `NeedlemanWunschScores`, `SmithWatermanScores`, `compute_into`;
//! bench.rs:79-82, :376-399, :466-487, :599-603, :658-670) so that the new rows read like the existing ones. On top
//! of the reference's dense cross-product every engine has the pairwise batch the backend was built for
//! (`pairs_into`), and tapes can be prepared once (`PreparedTape`) the way the reference builds its tape views
//! once outside the timed closures (bench.rs:292-306).
//!
//! Not compiled where it was written (no Rust toolchain there). `extern "C"` below is a one-to-one transcription
//! of the header; `tests/test_abi.py` checks that every symbol the header declares is declared here.
#![allow(clippy::too_many_arguments)]

use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr;

use stringtape::{BytesTapeView, CharsTapeView};

// ------------------------------------------------------------------------------------------------------------
// Raw ABI
// ------------------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct TapeU32 { pub data: *const u8, pub offsets: *const u32, pub count: usize }
#[repr(C)]
pub struct TapeU64 { pub data: *const u8, pub offsets: *const u64, pub count: usize }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PreparedView { pub tape: *mut c_void, pub first: usize, pub count: usize }
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PreparedInfo { pub count: usize, pub bytes: u64, pub symbols: u64, pub longest: u32, pub utf8: c_int, pub ascii: c_int }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct Timing {
    pub total_ms: f64, pub dominant_ms: f64, pub compute_ms: f64, pub dominant_name: [c_char; 64],
    pub cells: u64, pub bytes: u64, pub kernels: u32,
}
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct TimingTotals { pub total_ms: f64, pub dominant_ms: f64, pub compute_ms: f64, pub calls: u64 }
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ShardTiming { pub compute_ms: f64, pub gather_ms: f64, pub cells: u64, pub pairs: u64 }

pub const UNBOUNDED: u32 = u32::MAX;
pub const ALGORITHM_AUTO: c_int = 0;
pub const ALGORITHM_WAVEFRONT: c_int = 1;
pub const ALGORITHM_BITPARALLEL: c_int = 2;
pub const ALGORITHM_TILED: c_int = 3;

type Err = *mut *const c_char;
type Handle = *mut c_void;

#[rustfmt::skip]
extern "C" {
    // scopes
    fn swh_scope_init_gpu(device: c_int, scope: *mut Handle, error: Err) -> c_int;
    fn swh_scope_init_gpu_stream(device: c_int, hip_stream: *mut c_void, scope: *mut Handle, error: Err) -> c_int;
    fn swh_device_count(count: *mut c_int) -> c_int;
    fn swh_scope_init_gpus(devices: *const c_int, count: c_int, scope: *mut Handle, error: Err) -> c_int;
    fn swh_scope_init_cpu(cores: usize, scope: *mut Handle, error: Err) -> c_int;
    fn swh_scope_free(scope: Handle) -> c_int;
    fn swh_scope_compute_units(scope: Handle, compute_units: *mut usize) -> c_int;
    fn swh_scope_device_count(scope: Handle, devices: *mut usize) -> c_int;
    fn swh_scope_set_async(scope: Handle, enabled: c_int) -> c_int;
    fn swh_scope_synchronize(scope: Handle, error: Err) -> c_int;
    fn swh_scope_set_pipelined(scope: Handle, enabled: c_int, error: Err) -> c_int;
    fn swh_scope_join(scope: Handle, error: Err) -> c_int;
    fn swh_scope_forget(scope: Handle) -> c_int;
    fn swh_scope_describe(scope: Handle, text: *mut c_char, capacity: usize) -> c_int;
    fn swh_scope_set_profiling(scope: Handle, enabled: c_int) -> c_int;
    fn swh_scope_last_timing(scope: Handle, timing: *mut Timing) -> c_int;
    fn swh_scope_timing_totals(scope: Handle, totals: *mut TimingTotals) -> c_int;
    fn swh_scope_shard_timing(scope: Handle, timing: *mut ShardTiming) -> c_int;
    // memory
    fn swh_unified_alloc(scope: Handle, bytes: usize, pointer: *mut *mut c_void, error: Err) -> c_int;
    fn swh_unified_free(scope: Handle, pointer: *mut c_void) -> c_int;
    fn swh_device_alloc(scope: Handle, bytes: usize, pointer: *mut *mut c_void, error: Err) -> c_int;
    fn swh_device_free(scope: Handle, pointer: *mut c_void) -> c_int;
    fn swh_copy_to_device(scope: Handle, device_dst: *mut c_void, host_src: *const c_void, bytes: usize, error: Err) -> c_int;
    fn swh_copy_to_host(scope: Handle, host_dst: *mut c_void, device_src: *const c_void, bytes: usize, error: Err) -> c_int;
    // prepared tapes
    fn swh_tape_prepare_u32(scope: Handle, tape: *const TapeU32, utf8: c_int, prepared: *mut Handle, error: Err) -> c_int;
    fn swh_tape_prepare_u64(scope: Handle, tape: *const TapeU64, utf8: c_int, prepared: *mut Handle, error: Err) -> c_int;
    fn swh_prepared_info(prepared: Handle, info: *mut PreparedInfo) -> c_int;
    fn swh_prepared_free(prepared: Handle) -> c_int;
    // Levenshtein
    fn swh_levenshtein_init(scope: Handle, r#match: c_int, mismatch: c_int, open: c_int, extend: c_int, engine: *mut Handle, error: Err) -> c_int;
    fn swh_levenshtein_free(engine: Handle) -> c_int;
    fn swh_levenshtein_set_algorithm(engine: Handle, algorithm: c_int) -> c_int;
    fn swh_levenshtein_pairs_u32tape(engine: Handle, scope: Handle, a: *const TapeU32, b: *const TapeU32, bound: u32, out: *mut u32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_levenshtein_pairs_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, bound: u32, out: *mut u32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_levenshtein_utf8_pairs_u32tape(engine: Handle, scope: Handle, a: *const TapeU32, b: *const TapeU32, bound: u32, out: *mut u32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_levenshtein_utf8_pairs_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, bound: u32, out: *mut u32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_levenshtein_cross_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, out: *mut usize, row_stride_bytes: usize, error: Err) -> c_int;
    fn swh_levenshtein_utf8_cross_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, out: *mut usize, row_stride_bytes: usize, error: Err) -> c_int;
    fn swh_levenshtein_pairs_prepared(engine: Handle, scope: Handle, a: *const PreparedView, b: *const PreparedView, bound: u32, out: *mut u32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_levenshtein_cross_prepared(engine: Handle, scope: Handle, a: *const PreparedView, b: *const PreparedView, out: *mut usize, row_stride_bytes: usize, error: Err) -> c_int;
    fn swh_sharded_prepare_u32tape(scope: Handle, a: *const TapeU32, b: *const TapeU32, utf8: c_int, sharded: *mut Handle, error: Err) -> c_int;
    fn swh_sharded_prepare_u64tape(scope: Handle, a: *const TapeU64, b: *const TapeU64, utf8: c_int, sharded: *mut Handle, error: Err) -> c_int;
    fn swh_sharded_free(sharded: Handle) -> c_int;
    fn swh_sharded_cuts(sharded: Handle, cuts: *mut usize, capacity: usize) -> c_int;
    fn swh_levenshtein_pairs_sharded(engine: Handle, scope: Handle, sharded: Handle, bound: u32, out: *mut u32, error: Err) -> c_int;
    fn swh_sharded_cross_prepare_u64tape(scope: Handle, queries: *const TapeU64, candidates: *const TapeU64, utf8: c_int, product: *mut Handle, error: Err) -> c_int;
    fn swh_sharded_cross_free(product: Handle) -> c_int;
    fn swh_levenshtein_cross_sharded(engine: Handle, scope: Handle, product: Handle, matrix: *mut usize, row_stride_bytes: usize, error: Err) -> c_int;
    fn swh_nw_cross_sharded(engine: Handle, scope: Handle, product: Handle, matrix: *mut isize, row_stride_bytes: usize, error: Err) -> c_int;
    fn swh_sw_cross_sharded(engine: Handle, scope: Handle, product: Handle, matrix: *mut isize, row_stride_bytes: usize, error: Err) -> c_int;
    fn swh_nw_pairs_sharded(engine: Handle, scope: Handle, sharded: Handle, out: *mut i32, error: Err) -> c_int;
    fn swh_sw_pairs_sharded(engine: Handle, scope: Handle, sharded: Handle, out: *mut i32, error: Err) -> c_int;
    fn swh_levenshtein_pairs_sharded_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, bound: u32, out: *mut u32, error: Err) -> c_int;
    // Needleman-Wunsch
    fn swh_nw_init(scope: Handle, substitution_256x256: *const i8, open: c_int, extend: c_int, engine: *mut Handle, error: Err) -> c_int;
    fn swh_nw_init_classes(scope: Handle, byte_to_class_256: *const u8, class_costs_32x32: *const i8, open: c_int, extend: c_int, engine: *mut Handle, error: Err) -> c_int;
    fn swh_nw_free(engine: Handle) -> c_int;
    fn swh_nw_pairs_u32tape(engine: Handle, scope: Handle, a: *const TapeU32, b: *const TapeU32, out: *mut i32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_nw_pairs_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, out: *mut i32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_nw_cross_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, out: *mut isize, row_stride_bytes: usize, error: Err) -> c_int;
    fn swh_nw_pairs_prepared(engine: Handle, scope: Handle, a: *const PreparedView, b: *const PreparedView, out: *mut i32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_nw_cross_prepared(engine: Handle, scope: Handle, a: *const PreparedView, b: *const PreparedView, out: *mut isize, row_stride_bytes: usize, error: Err) -> c_int;
    // Smith-Waterman
    fn swh_sw_init(scope: Handle, substitution_256x256: *const i8, open: c_int, extend: c_int, engine: *mut Handle, error: Err) -> c_int;
    fn swh_sw_init_classes(scope: Handle, byte_to_class_256: *const u8, class_costs_32x32: *const i8, open: c_int, extend: c_int, engine: *mut Handle, error: Err) -> c_int;
    fn swh_sw_free(engine: Handle) -> c_int;
    fn swh_sw_pairs_u32tape(engine: Handle, scope: Handle, a: *const TapeU32, b: *const TapeU32, out: *mut i32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_sw_pairs_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, out: *mut i32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_sw_cross_u64tape(engine: Handle, scope: Handle, a: *const TapeU64, b: *const TapeU64, out: *mut isize, row_stride_bytes: usize, error: Err) -> c_int;
    fn swh_sw_pairs_prepared(engine: Handle, scope: Handle, a: *const PreparedView, b: *const PreparedView, out: *mut i32, out_stride_bytes: usize, error: Err) -> c_int;
    fn swh_sw_cross_prepared(engine: Handle, scope: Handle, a: *const PreparedView, b: *const PreparedView, out: *mut isize, row_stride_bytes: usize, error: Err) -> c_int;
    // introspection
    fn swh_version() -> *const c_char;
    fn swh_capabilities() -> *const c_char;
}

// ------------------------------------------------------------------------------------------------------------
// Errors: `E: Display`, as bench.rs:480-485 (`panic!("{}", error)`) and :632-635 (SKIPPED) need
// ------------------------------------------------------------------------------------------------------------
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Status { BadAlloc, InvalidArgument, InvalidUtf8, UnsupportedLength, NoDevice, DeviceError, NotImplemented, Other(i32) }
#[derive(Debug, Clone)]
pub struct Error { pub status: Status, pub message: String }
impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "{:?}: {}", self.status, self.message) }
}
impl std::error::Error for Error {}

fn check(status: c_int, message: *const c_char) -> Result<(), Error> {
    if status == 0 { return Ok(()); }
    let status = match status {
        1 => Status::BadAlloc, 2 => Status::InvalidArgument, 3 => Status::InvalidUtf8, 4 => Status::UnsupportedLength,
        5 => Status::NoDevice, 6 => Status::DeviceError, 7 => Status::NotImplemented, other => Status::Other(other),
    };
    let message = if message.is_null() { String::new() } else { unsafe { CStr::from_ptr(message) }.to_string_lossy().into_owned() };
    Err(Error { status, message })
}

/// HIP devices this process can see (for the `<Ngpu>` rows).
pub fn visible_devices() -> usize { let mut n: c_int = 0; unsafe { swh_device_count(&mut n) }; n.max(0) as usize }
pub fn version() -> String { unsafe { CStr::from_ptr(swh_version()) }.to_string_lossy().into_owned() }
/// What `log_stringzilla_metadata` prints for the other backends (utils.rs:78-92).
pub fn capabilities() -> String { unsafe { CStr::from_ptr(swh_capabilities()) }.to_string_lossy().into_owned() }

// ------------------------------------------------------------------------------------------------------------
// DeviceScope
// ------------------------------------------------------------------------------------------------------------
pub struct DeviceScope { handle: Handle }
unsafe impl Send for DeviceScope {}

impl DeviceScope {
    /// `DeviceScope::gpu_device(0)` (bench.rs:379, :652, :978).
    pub fn gpu_device(index: usize) -> Result<Self, Error> {
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_scope_init_gpu(index as c_int, &mut handle, &mut message) }, message)?;
        Ok(Self { handle })
    }
    /// Several GPUs of one node behind one scope: pair batches are split over them and the `u32` distances gathered
    /// with RCCL inside the library -- the `<8gpu>` rows.
    pub fn gpu_devices(indices: &[usize]) -> Result<Self, Error> {
        let devices: Vec<c_int> = indices.iter().map(|&i| i as c_int).collect();
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_scope_init_gpus(devices.as_ptr(), devices.len() as c_int, &mut handle, &mut message) }, message)?;
        Ok(Self { handle })
    }
    /// `DeviceScope::cpu_cores(n)` (bench.rs:376-378): this backend has no CPU path and says so (`NotImplemented`),
    /// which a harness turns into a SKIPPED line.
    pub fn cpu_cores(cores: usize) -> Result<Self, Error> {
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_scope_init_cpu(cores, &mut handle, &mut message) }, message)?;
        Ok(Self { handle })
    }
    /// Compute units, the "cores" of `auto_batch_size(gpu_multiprocessor_count(0)..)` (bench.rs:284-289).
    pub fn compute_units(&self) -> usize { let mut n = 0; unsafe { swh_scope_compute_units(self.handle, &mut n) }; n }
    pub fn device_count(&self) -> usize { let mut n = 0; unsafe { swh_scope_device_count(self.handle, &mut n) }; n }
    pub fn set_async(&self, enabled: bool) { unsafe { swh_scope_set_async(self.handle, enabled as c_int) }; }
    pub fn set_pipelined(&self, enabled: bool) -> Result<(), Error> {
        let mut message = ptr::null();
        check(unsafe { swh_scope_set_pipelined(self.handle, enabled as c_int, &mut message) }, message)
    }
    pub fn join(&self) -> Result<(), Error> { let mut m = ptr::null(); check(unsafe { swh_scope_join(self.handle, &mut m) }, m) }
    pub fn synchronize(&self) -> Result<(), Error> { let mut m = ptr::null(); check(unsafe { swh_scope_synchronize(self.handle, &mut m) }, m) }
    pub fn set_profiling(&self, enabled: bool) { unsafe { swh_scope_set_profiling(self.handle, enabled as c_int) }; }
    /// Drops what the scope believes about earlier calls: the next call is routed as on a new scope.
    pub fn forget(&self) { unsafe { swh_scope_forget(self.handle) }; }
    /// The scope's beliefs as one line of `key=value` pairs.
    pub fn describe(&self) -> String {
        let mut text = vec![0 as c_char; 512];
        unsafe { swh_scope_describe(self.handle, text.as_mut_ptr(), text.len()) };
        unsafe { std::ffi::CStr::from_ptr(text.as_ptr()) }.to_string_lossy().into_owned()
    }
    pub fn timing_totals(&self) -> TimingTotals { let mut t = TimingTotals::default(); unsafe { swh_scope_timing_totals(self.handle, &mut t) }; t }
    pub fn shard_timing(&self) -> ShardTiming { let mut t = ShardTiming::default(); unsafe { swh_scope_shard_timing(self.handle, &mut t) }; t }
    /// Cells and dominant kernel of the last call (`swh_scope_last_timing`), for the harness' own CUPS cross-check.
    pub fn last_timing(&self) -> (u64, f64, String) {
        let mut t = std::mem::MaybeUninit::<Timing>::zeroed();
        unsafe { swh_scope_last_timing(self.handle, t.as_mut_ptr()) };
        let t = unsafe { t.assume_init() };
        (t.cells, t.compute_ms, unsafe { CStr::from_ptr(t.dominant_name.as_ptr()) }.to_string_lossy().into_owned())
    }
    /// `UnifiedAlloc` parity (bench.rs:292-295): host-visible memory the device reads in place.
    pub fn unified_alloc(&self, bytes: usize) -> Result<*mut c_void, Error> {
        let (mut p, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_unified_alloc(self.handle, bytes, &mut p, &mut message) }, message)?;
        Ok(p)
    }
    pub fn unified_free(&self, pointer: *mut c_void) { unsafe { swh_unified_free(self.handle, pointer) }; }
    pub fn device_alloc(&self, bytes: usize) -> Result<*mut c_void, Error> {
        let (mut p, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_device_alloc(self.handle, bytes, &mut p, &mut message) }, message)?;
        Ok(p)
    }
    pub fn device_free(&self, pointer: *mut c_void) { unsafe { swh_device_free(self.handle, pointer) }; }
    pub fn copy_to_device(&self, device_dst: *mut c_void, host: &[u8]) -> Result<(), Error> {
        let mut message = ptr::null();
        check(unsafe { swh_copy_to_device(self.handle, device_dst, host.as_ptr() as *const c_void, host.len(), &mut message) }, message)
    }
    pub fn copy_to_host(&self, host: &mut [u8], device_src: *const c_void) -> Result<(), Error> {
        let mut message = ptr::null();
        check(unsafe { swh_copy_to_host(self.handle, host.as_mut_ptr() as *mut c_void, device_src, host.len(), &mut message) }, message)
    }
    /// `hipStream_t`-sharing constructor for callers that already own a stream.
    pub unsafe fn gpu_device_on_stream(index: usize, hip_stream: *mut c_void) -> Result<Self, Error> {
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(swh_scope_init_gpu_stream(index as c_int, hip_stream, &mut handle, &mut message), message)?;
        Ok(Self { handle })
    }
}
impl Drop for DeviceScope { fn drop(&mut self) { unsafe { swh_scope_free(self.handle) }; } }

// ------------------------------------------------------------------------------------------------------------
// Tapes
// ------------------------------------------------------------------------------------------------------------
fn bytes_tape(view: &BytesTapeView<u64>) -> TapeU64 { TapeU64 { data: view.data().as_ptr(), offsets: view.offsets().as_ptr(), count: view.len() } }
fn chars_tape(view: &CharsTapeView<u64>) -> TapeU64 { TapeU64 { data: view.data().as_ptr(), offsets: view.offsets().as_ptr(), count: view.len() } }

/// A tape made ready once: resident on the device, measured, and -- for chars -- validated and decoded
/// (`swh_tape_prepare_*`). The lifetime ties a prepared tape to the view it may borrow device memory from.
pub struct PreparedTape<'tape> { handle: Handle, first: usize, count: usize, owner: bool, _borrow: PhantomData<&'tape [u8]> }
impl<'tape> PreparedTape<'tape> {
    pub fn bytes(scope: &DeviceScope, view: &'tape BytesTapeView<u64>) -> Result<Self, Error> {
        let tape = bytes_tape(view);
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_tape_prepare_u64(scope.handle, &tape, 0, &mut handle, &mut message) }, message)?;
        Ok(Self { handle, first: 0, count: view.len(), owner: true, _borrow: PhantomData })
    }
    /// Fails with `InvalidUtf8` where `CharsTapeView::try_from` would (bench.rs:303-306); a `CharsTapeView` is valid already.
    pub fn chars(scope: &DeviceScope, view: &'tape CharsTapeView<u64>) -> Result<Self, Error> {
        let tape = chars_tape(view);
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_tape_prepare_u64(scope.handle, &tape, 1, &mut handle, &mut message) }, message)?;
        Ok(Self { handle, first: 0, count: view.len(), owner: true, _borrow: PhantomData })
    }
    /// 32-bit offsets halve the offset traffic of short-word shards.
    pub fn bytes_u32(scope: &DeviceScope, data: &'tape [u8], offsets: &'tape [u32]) -> Result<Self, Error> {
        let tape = TapeU32 { data: data.as_ptr(), offsets: offsets.as_ptr(), count: offsets.len().saturating_sub(1) };
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_tape_prepare_u32(scope.handle, &tape, 0, &mut handle, &mut message) }, message)?;
        Ok(Self { handle, first: 0, count: tape.count, owner: true, _borrow: PhantomData })
    }
    /// `BytesTapeView::subview(lo, hi)` (bench.rs:134-139): zero-copy.
    pub fn subview(&self, lo: usize, hi: usize) -> Option<PreparedTape<'_>> {
        if lo > hi || hi > self.count { return None; }
        Some(PreparedTape { handle: self.handle, first: self.first + lo, count: hi - lo, owner: false, _borrow: PhantomData })
    }
    pub fn len(&self) -> usize { self.count }
    pub fn is_empty(&self) -> bool { self.count == 0 }
    pub fn info(&self) -> PreparedInfo { let mut info = PreparedInfo::default(); unsafe { swh_prepared_info(self.handle, &mut info) }; info }
    fn view(&self) -> PreparedView { PreparedView { tape: self.handle, first: self.first, count: self.count } }
}
impl Drop for PreparedTape<'_> { fn drop(&mut self) { if self.owner { unsafe { swh_prepared_free(self.handle) }; } } }

/// A pairwise batch made resident on every device of a multi-GPU scope (`swh_sharded_prepare_*`): contiguous shards
/// balanced on DP cells, shard r uploaded to and prepared on device r. The steady state of the `<Ngpu>` rows.
pub struct ShardedPairs { handle: Handle, pairs: usize }
impl ShardedPairs {
    pub fn bytes(scope: &DeviceScope, a: &BytesTapeView<u64>, b: &BytesTapeView<u64>) -> Result<Self, Error> {
        let (ta, tb) = (bytes_tape(a), bytes_tape(b));
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_sharded_prepare_u64tape(scope.handle, &ta, &tb, 0, &mut handle, &mut message) }, message)?;
        Ok(Self { handle, pairs: a.len() })
    }
    pub fn bytes_u32(scope: &DeviceScope, a: &TapeU32, b: &TapeU32) -> Result<Self, Error> {
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_sharded_prepare_u32tape(scope.handle, a, b, 0, &mut handle, &mut message) }, message)?;
        Ok(Self { handle, pairs: a.count })
    }
    pub fn len(&self) -> usize { self.pairs }
    pub fn is_empty(&self) -> bool { self.pairs == 0 }
    /// Pair indices where the shards begin / end (`devices + 1` entries).
    pub fn cuts(&self, devices: usize) -> Vec<usize> {
        let mut cuts = vec![0usize; devices + 1];
        unsafe { swh_sharded_cuts(self.handle, cuts.as_mut_ptr(), cuts.len()) };
        cuts
    }
}
impl Drop for ShardedPairs { fn drop(&mut self) { unsafe { swh_sharded_free(self.handle) }; } }

/// A dense queries x candidates product made resident on every device of a multi-GPU scope
/// (`swh_sharded_cross_prepare_u64tape`): row blocks of equal query symbols, block r and all candidates prepared on device r.
/// `compute_into_sharded` is the `<Ngpu>` twin of `compute_into` (bench.rs:478-486).
pub struct ShardedCross { handle: Handle, rows: usize, columns: usize }
impl ShardedCross {
    pub fn bytes(scope: &DeviceScope, queries: &BytesTapeView<u64>, candidates: &BytesTapeView<u64>) -> Result<Self, Error> {
        let (tq, tc) = (bytes_tape(queries), bytes_tape(candidates));
        let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
        check(unsafe { swh_sharded_cross_prepare_u64tape(scope.handle, &tq, &tc, 0, &mut handle, &mut message) }, message)?;
        Ok(Self { handle, rows: queries.len(), columns: candidates.len() })
    }
    pub fn shape(&self) -> (usize, usize) { (self.rows, self.columns) }
}
impl Drop for ShardedCross { fn drop(&mut self) { unsafe { swh_sharded_cross_free(self.handle) }; } }

// ------------------------------------------------------------------------------------------------------------
// Engines
// ------------------------------------------------------------------------------------------------------------
pub struct LevenshteinDistances { handle: Handle }
pub struct LevenshteinDistancesUtf8 { handle: Handle }

fn levenshtein_engine(scope: &DeviceScope, m: i32, x: i32, o: i32, e: i32) -> Result<Handle, Error> {
    let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
    check(unsafe { swh_levenshtein_init(scope.handle, m, x, o, e, &mut handle, &mut message) }, message)?;
    Ok(handle)
}

impl LevenshteinDistances {
    /// `LevenshteinDistances::new(&scope, 0, 1, 1, 1)` (bench.rs:382-385, :390-393).
    pub fn new(scope: &DeviceScope, match_cost: i32, mismatch: i32, open: i32, extend: i32) -> Result<Self, Error> {
        Ok(Self { handle: levenshtein_engine(scope, match_cost, mismatch, open, extend)? })
    }
    pub fn set_algorithm(&self, algorithm: c_int) { unsafe { swh_levenshtein_set_algorithm(self.handle, algorithm) }; }
    /// Pairwise batch: `out[i] = min(d(a_i, b_i), bound + 1)`; `None` = unbounded. The batched form of the loop
    /// `rapidfuzz::distance::levenshtein::distance(a.iter().copied(), b.iter().copied())` (bench.rs:404-423).
    pub fn pairs_into(&self, scope: &DeviceScope, a: &BytesTapeView<u64>, b: &BytesTapeView<u64>, bound: Option<u32>, out: &mut [u32]) -> Result<(), Error> {
        assert!(out.len() >= a.len(), "one result per pair");
        let (ta, tb) = (bytes_tape(a), bytes_tape(b));
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_pairs_u64tape(self.handle, scope.handle, &ta, &tb, bound.unwrap_or(UNBOUNDED), out.as_mut_ptr(), 4, &mut message) }, message)?;
        #[cfg(feature = "verify-rapidfuzz")]
        verify_against_rapidfuzz((0..a.len()).map(|i| (&a[i], &b[i])), bound, out);
        Ok(())
    }
    pub fn pairs_into_u32(&self, scope: &DeviceScope, a: &TapeU32, b: &TapeU32, bound: Option<u32>, out: &mut [u32]) -> Result<(), Error> {
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_pairs_u32tape(self.handle, scope.handle, a, b, bound.unwrap_or(UNBOUNDED), out.as_mut_ptr(), 4, &mut message) }, message)
    }
    /// The same on tapes prepared once (no per-call planning pre-pass; sub-views are free).
    pub fn pairs_into_prepared(&self, scope: &DeviceScope, a: &PreparedTape, b: &PreparedTape, bound: Option<u32>, out: &mut [u32]) -> Result<(), Error> {
        let (va, vb) = (a.view(), b.view());
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_pairs_prepared(self.handle, scope.handle, &va, &vb, bound.unwrap_or(UNBOUNDED), out.as_mut_ptr(), 4, &mut message) }, message)
    }
    /// A resident sharded batch over every GPU of a multi-device scope, distances gathered with RCCL (`<Ngpu>` rows).
    pub fn pairs_into_sharded_resident(&self, scope: &DeviceScope, batch: &ShardedPairs, bound: Option<u32>, out: &mut [u32]) -> Result<(), Error> {
        assert!(out.len() >= batch.len());
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_pairs_sharded(self.handle, scope.handle, batch.handle, bound.unwrap_or(UNBOUNDED), out.as_mut_ptr(), &mut message) }, message)
    }
    /// The dense matrix of a sharded product: every device fills its rows and copies them into `matrix`.
    pub fn compute_into_sharded(&self, scope: &DeviceScope, product: &ShardedCross, matrix: &mut [usize]) -> Result<(), Error> {
        let (rows, columns) = product.shape();
        assert!(matrix.len() >= rows * columns);
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_cross_sharded(self.handle, scope.handle, product.handle, matrix.as_mut_ptr(), columns * 8, &mut message) }, message)
    }
    /// One-shot form: shard, upload, score, gather, free.
    pub fn pairs_into_sharded(&self, scope: &DeviceScope, a: &BytesTapeView<u64>, b: &BytesTapeView<u64>, bound: Option<u32>, out: &mut [u32]) -> Result<(), Error> {
        let (ta, tb) = (bytes_tape(a), bytes_tape(b));
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_pairs_sharded_u64tape(self.handle, scope.handle, &ta, &tb, bound.unwrap_or(UNBOUNDED), out.as_mut_ptr(), &mut message) }, message)
    }
    /// `compute_into(&scope, AnyBytesTape::View64(q), Some(AnyBytesTape::View64(c)), &mut matrix)` (bench.rs:478-486,
    /// :599-603): dense `q.len() x c.len()` row-major `usize`; `None` = q x q.
    pub fn compute_into(&self, scope: &DeviceScope, queries: &BytesTapeView<u64>, candidates: Option<&BytesTapeView<u64>>, matrix: &mut [usize]) -> Result<(), Error> {
        let columns = candidates.map_or(queries.len(), |c| c.len());
        assert!(matrix.len() >= queries.len() * columns);
        let tq = bytes_tape(queries);
        let tc = candidates.map(bytes_tape);
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_cross_u64tape(self.handle, scope.handle, &tq, tc.as_ref().map_or(ptr::null(), |t| t as *const TapeU64),
                                                     matrix.as_mut_ptr(), columns * 8, &mut message) }, message)
    }
    pub fn compute_into_prepared(&self, scope: &DeviceScope, queries: &PreparedTape, candidates: Option<&PreparedTape>, matrix: &mut [usize]) -> Result<(), Error> {
        let columns = candidates.map_or(queries.len(), |c| c.len());
        let vq = queries.view();
        let vc = candidates.map(|c| c.view());
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_cross_prepared(self.handle, scope.handle, &vq, vc.as_ref().map_or(ptr::null(), |v| v as *const PreparedView),
                                                      matrix.as_mut_ptr(), columns * 8, &mut message) }, message)
    }
    /// The allocating form, `engine.compute(&scope, &queries, &candidates)` (bench.rs:466-468).
    pub fn compute(&self, scope: &DeviceScope, queries: &BytesTapeView<u64>, candidates: &BytesTapeView<u64>) -> Result<Vec<usize>, Error> {
        let mut matrix = vec![0usize; queries.len() * candidates.len()];
        self.compute_into(scope, queries, Some(candidates), &mut matrix)?;
        Ok(matrix)
    }
}
impl Drop for LevenshteinDistances { fn drop(&mut self) { unsafe { swh_levenshtein_free(self.handle) }; } }

impl LevenshteinDistancesUtf8 {
    /// `LevenshteinDistancesUtf8::new(&scope, 0, 1, 1, 1)` (bench.rs:386-389, :396-399): symbols are `char`s.
    pub fn new(scope: &DeviceScope, match_cost: i32, mismatch: i32, open: i32, extend: i32) -> Result<Self, Error> {
        Ok(Self { handle: levenshtein_engine(scope, match_cost, mismatch, open, extend)? })
    }
    pub fn pairs_into(&self, scope: &DeviceScope, a: &CharsTapeView<u64>, b: &CharsTapeView<u64>, bound: Option<u32>, out: &mut [u32]) -> Result<(), Error> {
        let (ta, tb) = (chars_tape(a), chars_tape(b));
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_utf8_pairs_u64tape(self.handle, scope.handle, &ta, &tb, bound.unwrap_or(UNBOUNDED), out.as_mut_ptr(), 4, &mut message) }, message)
    }
    pub fn pairs_into_u32(&self, scope: &DeviceScope, a: &TapeU32, b: &TapeU32, bound: Option<u32>, out: &mut [u32]) -> Result<(), Error> {
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_utf8_pairs_u32tape(self.handle, scope.handle, a, b, bound.unwrap_or(UNBOUNDED), out.as_mut_ptr(), 4, &mut message) }, message)
    }
    pub fn pairs_into_prepared(&self, scope: &DeviceScope, a: &PreparedTape, b: &PreparedTape, bound: Option<u32>, out: &mut [u32]) -> Result<(), Error> {
        let (va, vb) = (a.view(), b.view());
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_pairs_prepared(self.handle, scope.handle, &va, &vb, bound.unwrap_or(UNBOUNDED), out.as_mut_ptr(), 4, &mut message) }, message)
    }
    /// `compute_into(&scope, AnyCharsTape::View64(q), Some(AnyCharsTape::View64(c)), &mut matrix)` (bench.rs:538-546, :625-629).
    pub fn compute_into(&self, scope: &DeviceScope, queries: &CharsTapeView<u64>, candidates: Option<&CharsTapeView<u64>>, matrix: &mut [usize]) -> Result<(), Error> {
        let columns = candidates.map_or(queries.len(), |c| c.len());
        assert!(matrix.len() >= queries.len() * columns);
        let tq = chars_tape(queries);
        let tc = candidates.map(chars_tape);
        let mut message = ptr::null();
        check(unsafe { swh_levenshtein_utf8_cross_u64tape(self.handle, scope.handle, &tq, tc.as_ref().map_or(ptr::null(), |t| t as *const TapeU64),
                                                          matrix.as_mut_ptr(), columns * 8, &mut message) }, message)
    }
}
impl Drop for LevenshteinDistancesUtf8 { fn drop(&mut self) { unsafe { swh_levenshtein_free(self.handle) }; } }

/// Alignment engines share their plumbing; `$init*`/`$pairs*`/`$cross*` pick the NW or SW entry points.
macro_rules! alignment_engine {
    ($name:ident, $doc:expr, $init:ident, $init_classes:ident, $free:ident, $pairs32:ident, $pairs64:ident, $cross:ident, $pairs_prepared:ident, $cross_prepared:ident, $pairs_sharded:ident, $cross_sharded:ident) => {
        #[doc = $doc]
        pub struct $name { handle: Handle }
        impl $name {
            /// `::new(&scope, &byte_to_class, &class_costs, open, extend)` (bench.rs:658-662, :667-670, :985-997):
            /// 32 symbol classes; gap(k) = open + (k-1)*extend.
            pub fn new(scope: &DeviceScope, byte_to_class: &[u8; 256], class_costs: &[[i8; 32]; 32], open: i8, extend: i8) -> Result<Self, Error> {
                let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
                check(unsafe { $init_classes(scope.handle, byte_to_class.as_ptr(), class_costs.as_ptr() as *const i8, open as c_int, extend as c_int, &mut handle, &mut message) }, message)?;
                Ok(Self { handle })
            }
            /// Full 256x256 `i8` substitution matrix (rust-bio style scoring closures tabulated; BASELINE config C4).
            pub fn with_matrix(scope: &DeviceScope, substitution: &[[i8; 256]; 256], open: i32, extend: i32) -> Result<Self, Error> {
                let (mut handle, mut message) = (ptr::null_mut(), ptr::null());
                check(unsafe { $init(scope.handle, substitution.as_ptr() as *const i8, open, extend, &mut handle, &mut message) }, message)?;
                Ok(Self { handle })
            }
            /// Pairwise batch of scores: the batched form of `Aligner::global(a, b).score` / `.local` (bench.rs:746-791).
            pub fn pairs_into(&self, scope: &DeviceScope, a: &BytesTapeView<u64>, b: &BytesTapeView<u64>, out: &mut [i32]) -> Result<(), Error> {
                let (ta, tb) = (bytes_tape(a), bytes_tape(b));
                let mut message = ptr::null();
                check(unsafe { $pairs64(self.handle, scope.handle, &ta, &tb, out.as_mut_ptr(), 4, &mut message) }, message)
            }
            pub fn pairs_into_u32(&self, scope: &DeviceScope, a: &TapeU32, b: &TapeU32, out: &mut [i32]) -> Result<(), Error> {
                let mut message = ptr::null();
                check(unsafe { $pairs32(self.handle, scope.handle, a, b, out.as_mut_ptr(), 4, &mut message) }, message)
            }
            pub fn pairs_into_prepared(&self, scope: &DeviceScope, a: &PreparedTape, b: &PreparedTape, out: &mut [i32]) -> Result<(), Error> {
                let (va, vb) = (a.view(), b.view());
                let mut message = ptr::null();
                check(unsafe { $pairs_prepared(self.handle, scope.handle, &va, &vb, out.as_mut_ptr(), 4, &mut message) }, message)
            }
            /// One batch over every GPU of a multi-device scope: the matrix is cloned to each device on first use, the scores
            /// are gathered inside the library (RCCL) in pair order.
            pub fn pairs_into_sharded_resident(&self, scope: &DeviceScope, batch: &ShardedPairs, out: &mut [i32]) -> Result<(), Error> {
                assert!(out.len() >= batch.len());
                let mut message = ptr::null();
                check(unsafe { $pairs_sharded(self.handle, scope.handle, batch.handle, out.as_mut_ptr(), &mut message) }, message)
            }
            pub fn compute_into_sharded(&self, scope: &DeviceScope, product: &ShardedCross, matrix: &mut [isize]) -> Result<(), Error> {
                let (rows, columns) = product.shape();
                assert!(matrix.len() >= rows * columns);
                let mut message = ptr::null();
                check(unsafe { $cross_sharded(self.handle, scope.handle, product.handle, matrix.as_mut_ptr(), columns * 8, &mut message) }, message)
            }
            /// `compute_into(..) -> UnifiedMat<isize>` (bench.rs:814-821, :872-876).
            pub fn compute_into(&self, scope: &DeviceScope, queries: &BytesTapeView<u64>, candidates: Option<&BytesTapeView<u64>>, matrix: &mut [isize]) -> Result<(), Error> {
                let columns = candidates.map_or(queries.len(), |c| c.len());
                assert!(matrix.len() >= queries.len() * columns);
                let tq = bytes_tape(queries);
                let tc = candidates.map(bytes_tape);
                let mut message = ptr::null();
                check(unsafe { $cross(self.handle, scope.handle, &tq, tc.as_ref().map_or(ptr::null(), |t| t as *const TapeU64), matrix.as_mut_ptr(), columns * 8, &mut message) }, message)
            }
            pub fn compute_into_prepared(&self, scope: &DeviceScope, queries: &PreparedTape, candidates: Option<&PreparedTape>, matrix: &mut [isize]) -> Result<(), Error> {
                let columns = candidates.map_or(queries.len(), |c| c.len());
                let vq = queries.view();
                let vc = candidates.map(|c| c.view());
                let mut message = ptr::null();
                check(unsafe { $cross_prepared(self.handle, scope.handle, &vq, vc.as_ref().map_or(ptr::null(), |v| v as *const PreparedView), matrix.as_mut_ptr(), columns * 8, &mut message) }, message)
            }
        }
        impl Drop for $name { fn drop(&mut self) { unsafe { $free(self.handle) }; } }
    };
}
alignment_engine!(NeedlemanWunschScores, "`szs::NeedlemanWunschScores` (bench.rs:658-670): global alignment scores, linear or affine gaps.",
                  swh_nw_init, swh_nw_init_classes, swh_nw_free, swh_nw_pairs_u32tape, swh_nw_pairs_u64tape, swh_nw_cross_u64tape,
                  swh_nw_pairs_prepared, swh_nw_cross_prepared, swh_nw_pairs_sharded, swh_nw_cross_sharded);
alignment_engine!(SmithWatermanScores, "`szs::SmithWatermanScores` (bench.rs:882-963): local alignment scores.",
                  swh_sw_init, swh_sw_init_classes, swh_sw_free, swh_sw_pairs_u32tape, swh_sw_pairs_u64tape, swh_sw_cross_u64tape,
                  swh_sw_pairs_prepared, swh_sw_cross_prepared, swh_sw_pairs_sharded, swh_sw_cross_sharded);

/// The only route by which parity with the reference's own oracle can be pinned: with `--features verify-rapidfuzz`
/// every distance of every pairwise call is compared with `rapidfuzz::distance::levenshtein::distance` on the same
/// pair (bench.rs:416-419), honouring the cutoff convention `min(d, bound + 1)`.
#[cfg(feature = "verify-rapidfuzz")]
fn verify_against_rapidfuzz<'a>(pairs: impl Iterator<Item = (&'a [u8], &'a [u8])>, bound: Option<u32>, out: &[u32]) {
    for (index, (a, b)) in pairs.enumerate() {
        let exact = rapidfuzz::distance::levenshtein::distance(a.iter().copied(), b.iter().copied()) as u32;
        let expected = match bound { Some(k) if exact > k => k + 1, _ => exact };
        assert_eq!(out[index], expected, "stringwars_amd disagrees with rapidfuzz on pair {}", index);
    }
}
