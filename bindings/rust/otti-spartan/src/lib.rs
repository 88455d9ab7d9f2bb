//! libspartan's interface (`Instance`, `VarsAssignment`, `InputsAssignment`, `NIZKGens`, `NIZK::{prove, verify}`, and SNARK mode's `SNARKGens`,
//! `ComputationCommitment`, `SNARK::{encode, prove, verify}`) on top of the
//! MI355X prover's C ABI (`include/otti_spartan.h`).  Signatures follow upstream `src/lib.rs` so that a caller only changes its
//! `use` line; the one visible difference is the transcript: upstream takes `&mut merlin::Transcript`, this crate needs the label
//! the caller created it with (`Transcript::new(label)`), because the Fiat-Shamir transcript of the proof lives inside the library.
//! Proving needs a gfx950 device (status -20 otherwise); verification runs on the host.
use std::os::raw::{c_char, c_void};
use std::ptr;

#[repr(C)]
pub struct OttiEntry {
    pub row: u64,
    pub col: u64,
    pub val: [u8; 32],
}
pub enum OttiInstance {}
pub enum OttiGens {}
pub enum OttiSnarkGens {}
pub enum OttiCompComm {}

extern "C" {
    fn otti_instance_new(num_cons: u64, num_vars: u64, num_inputs: u64, a: *const OttiEntry, na: usize, b: *const OttiEntry, nb: usize,
                         c: *const OttiEntry, nc: usize, out: *mut *mut OttiInstance) -> i32;
    fn otti_instance_free(inst: *mut OttiInstance);
    fn otti_instance_is_sat(inst: *const OttiInstance, vars32: *const u8, nvars: usize, inputs32: *const u8, ninputs: usize, sat: *mut i32) -> i32;
    fn otti_gens_new(num_cons: u64, num_vars: u64, num_inputs: u64, out: *mut *mut OttiGens) -> i32;
    fn otti_gens_free(gens: *mut OttiGens);
    fn otti_nizk_prove(inst: *mut OttiInstance, vars32: *const u8, nvars: usize, inputs32: *const u8, ninputs: usize, gens: *mut OttiGens,
                       tlabel: *const u8, tlabel_len: usize, seed32: *const u8, flags: u32, proof: *mut *mut u8, proof_len: *mut usize,
                       stage_ms: *mut f64) -> i32;
    fn otti_nizk_verify(inst: *const OttiInstance, inputs32: *const u8, ninputs: usize, gens: *const OttiGens, tlabel: *const u8,
                        tlabel_len: usize, proof: *const u8, proof_len: usize) -> i32;
    // SNARK mode (upstream spartan-zkinterface without --nizk)
    fn otti_snark_gens_new(num_cons: u64, num_vars: u64, num_inputs: u64, num_nz_entries: u64, out: *mut *mut OttiSnarkGens) -> i32;
    fn otti_snark_gens_free(gens: *mut OttiSnarkGens);
    fn otti_snark_encode(inst: *mut OttiInstance, gens: *mut OttiSnarkGens, out: *mut *mut OttiCompComm) -> i32;
    fn otti_comp_comm_bytes(comm: *const OttiCompComm, out: *mut *mut u8, len: *mut usize) -> i32;
    fn otti_comp_comm_from_bytes(buf: *const u8, len: usize, out: *mut *mut OttiCompComm) -> i32;
    fn otti_comp_comm_free(comm: *mut OttiCompComm);
    fn otti_snark_prove(inst: *mut OttiInstance, comm: *mut OttiCompComm, vars32: *const u8, nvars: usize, inputs32: *const u8, ninputs: usize,
                        gens: *mut OttiSnarkGens, tlabel: *const u8, tlabel_len: usize, seed32: *const u8, flags: u32, proof: *mut *mut u8,
                        proof_len: *mut usize, stage_ms: *mut f64) -> i32;
    fn otti_snark_verify(comm: *const OttiCompComm, inputs32: *const u8, ninputs: usize, gens: *const OttiSnarkGens, tlabel: *const u8,
                         tlabel_len: usize, proof: *const u8, proof_len: usize) -> i32;
    fn otti_buf_free(p: *mut c_void);
    fn otti_last_error(buf: *mut c_char, cap: usize) -> usize;
}

const OTTI_FLAG_GPU: u32 = 1;

/// upstream `errors.rs`
#[derive(Debug, PartialEq, Eq, Clone, Copy)]
pub enum R1CSError {
    NonPowerOfTwoCons,
    NonPowerOfTwoVars,
    InvalidNumberOfInputs,
    InvalidNumberOfVars,
    InvalidScalar,
    InvalidIndex,
}
#[derive(Debug, PartialEq, Eq, Clone, Copy)]
pub enum ProofVerifyError {
    InternalError,
    DecompressionError([u8; 32]),
}

fn r1cs_error(rc: i32) -> R1CSError {
    match rc {
        -1 => R1CSError::NonPowerOfTwoCons,
        -2 => R1CSError::NonPowerOfTwoVars,
        -3 => R1CSError::InvalidNumberOfInputs,
        -4 => R1CSError::InvalidNumberOfVars,
        -5 => R1CSError::InvalidScalar,
        _ => R1CSError::InvalidIndex,
    }
}

/// the calling thread's last error message from the library
pub fn last_error() -> String {
    let mut buf = vec![0u8; 512];
    let n = unsafe { otti_last_error(buf.as_mut_ptr() as *mut c_char, buf.len()) };
    buf.truncate(n.min(511));
    String::from_utf8_lossy(&buf).into_owned()
}

fn canonical(s: &[u8; 32]) -> bool {
    // l = 2^252 + 27742317777372353535851937790883648493, little-endian
    const L: [u8; 32] = [0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                         0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10];
    for i in (0..32).rev() {
        if s[i] < L[i] { return true; }
        if s[i] > L[i] { return false; }
    }
    false
}

pub struct Instance { h: *mut OttiInstance }
unsafe impl Send for Instance {}
unsafe impl Sync for Instance {} // read-only after construction; see the threading note in otti_spartan.h
impl Drop for Instance { fn drop(&mut self) { unsafe { otti_instance_free(self.h) } } }

impl Instance {
    pub fn new(num_cons: usize, num_vars: usize, num_inputs: usize, a: &[(usize, usize, [u8; 32])], b: &[(usize, usize, [u8; 32])],
               c: &[(usize, usize, [u8; 32])]) -> Result<Instance, R1CSError> {
        let cv = |m: &[(usize, usize, [u8; 32])]| -> Vec<OttiEntry> {
            m.iter().map(|&(r, c, v)| OttiEntry { row: r as u64, col: c as u64, val: v }).collect()
        };
        let (a, b, c) = (cv(a), cv(b), cv(c));
        let mut h = ptr::null_mut();
        let rc = unsafe {
            otti_instance_new(num_cons as u64, num_vars as u64, num_inputs as u64, a.as_ptr(), a.len(), b.as_ptr(), b.len(), c.as_ptr(), c.len(), &mut h)
        };
        if rc == 0 { Ok(Instance { h }) } else { Err(r1cs_error(rc)) }
    }

    pub fn is_sat(&self, vars: &VarsAssignment, inputs: &InputsAssignment) -> Result<bool, R1CSError> {
        let mut sat = 0i32;
        let rc = unsafe {
            otti_instance_is_sat(self.h, vars.bytes.as_ptr() as *const u8, vars.bytes.len(), inputs.bytes.as_ptr() as *const u8, inputs.bytes.len(), &mut sat)
        };
        if rc == 0 { Ok(sat != 0) } else { Err(r1cs_error(rc)) }
    }
}

#[derive(Clone)]
pub struct Assignment { bytes: Vec<[u8; 32]> }
pub type VarsAssignment = Assignment;
pub type InputsAssignment = Assignment;
impl Assignment {
    pub fn new(assignment: &[[u8; 32]]) -> Result<Assignment, R1CSError> {
        if assignment.iter().all(canonical) { Ok(Assignment { bytes: assignment.to_vec() }) } else { Err(R1CSError::InvalidScalar) }
    }
}

pub struct NIZKGens { h: *mut OttiGens }
unsafe impl Send for NIZKGens {}
unsafe impl Sync for NIZKGens {}
impl Drop for NIZKGens { fn drop(&mut self) { unsafe { otti_gens_free(self.h) } } }
impl NIZKGens {
    pub fn new(num_cons: usize, num_vars: usize, num_inputs: usize) -> NIZKGens {
        let mut h = ptr::null_mut();
        let rc = unsafe { otti_gens_new(num_cons as u64, num_vars as u64, num_inputs as u64, &mut h) };
        assert_eq!(rc, 0, "otti_gens_new: {}", last_error());
        NIZKGens { h }
    }
}

/// a proof, bincode-compatible bytes of upstream's `NIZK` (see DESIGN.md section 0 for the compatibility caveat)
pub struct NIZK { pub bytes: Vec<u8> }

impl NIZK {
    /// `transcript_label`: what the caller would have passed to `merlin::Transcript::new`.
    pub fn prove(inst: &Instance, vars: VarsAssignment, inputs: &InputsAssignment, gens: &NIZKGens, transcript_label: &'static [u8]) -> NIZK {
        let (mut p, mut n) = (ptr::null_mut::<u8>(), 0usize);
        let rc = unsafe {
            otti_nizk_prove(inst.h, vars.bytes.as_ptr() as *const u8, vars.bytes.len(), inputs.bytes.as_ptr() as *const u8, inputs.bytes.len(), gens.h,
                            transcript_label.as_ptr(), transcript_label.len(), ptr::null(), OTTI_FLAG_GPU, &mut p, &mut n, ptr::null_mut())
        };
        assert_eq!(rc, 0, "otti_nizk_prove: {}", last_error()); // upstream's prove panics on malformed input as well
        let bytes = unsafe { std::slice::from_raw_parts(p, n) }.to_vec();
        unsafe { otti_buf_free(p as *mut c_void) };
        NIZK { bytes }
    }

    pub fn verify(&self, inst: &Instance, inputs: &InputsAssignment, transcript_label: &'static [u8], gens: &NIZKGens) -> Result<(), ProofVerifyError> {
        let rc = unsafe {
            otti_nizk_verify(inst.h, inputs.bytes.as_ptr() as *const u8, inputs.bytes.len(), gens.h, transcript_label.as_ptr(), transcript_label.len(),
                             self.bytes.as_ptr(), self.bytes.len())
        };
        match rc {
            0 => Ok(()),
            -11 => Err(ProofVerifyError::DecompressionError([0u8; 32])), // the C ABI does not report which point failed to decompress
            _ => Err(ProofVerifyError::InternalError),
        }
    }
}

// ------------------------------------------------------------------------------------------------ SNARK mode
// upstream lib.rs: SNARKGens::new(num_cons, num_vars, num_inputs, num_nz_entries); SNARK::encode(&inst, &gens) -> (ComputationCommitment,
// ComputationDecommitment); SNARK::prove(&inst, &comm, &decomm, vars, &inputs, &gens, &mut transcript); proof.verify(&comm, &inputs, &mut
// transcript, &gens).  The decommitment lives inside the handle `encode` returns (it is tens of MB of HBM-resident tables), so
// `ComputationDecommitment` here is a marker that borrows it.  bindings/c/otti_caller.c (`snark`, `snark-host`) is the compiled twin.

pub struct SNARKGens { h: *mut OttiSnarkGens }
unsafe impl Send for SNARKGens {}
unsafe impl Sync for SNARKGens {}
impl Drop for SNARKGens { fn drop(&mut self) { unsafe { otti_snark_gens_free(self.h) } } }
impl SNARKGens {
    pub fn new(num_cons: usize, num_vars: usize, num_inputs: usize, num_nz_entries: usize) -> SNARKGens {
        let mut h = ptr::null_mut();
        let rc = unsafe { otti_snark_gens_new(num_cons as u64, num_vars as u64, num_inputs as u64, num_nz_entries as u64, &mut h) };
        assert_eq!(rc, 0, "otti_snark_gens_new: {}", last_error());
        SNARKGens { h }
    }
}

pub struct ComputationCommitment { h: *mut OttiCompComm }
unsafe impl Send for ComputationCommitment {}
impl Drop for ComputationCommitment { fn drop(&mut self) { unsafe { otti_comp_comm_free(self.h) } } }
/// what `SNARK::encode` returns beside the commitment: the prover's side of it, kept inside the commitment's handle
pub struct ComputationDecommitment { _private: () }
impl ComputationCommitment {
    /// bincode of upstream's `ComputationCommitment` (what a verifier stores)
    pub fn to_bytes(&self) -> Vec<u8> {
        let (mut p, mut n) = (ptr::null_mut::<u8>(), 0usize);
        let rc = unsafe { otti_comp_comm_bytes(self.h, &mut p, &mut n) };
        assert_eq!(rc, 0, "otti_comp_comm_bytes: {}", last_error());
        let v = unsafe { std::slice::from_raw_parts(p, n) }.to_vec();
        unsafe { otti_buf_free(p as *mut c_void) };
        v
    }
    /// the verifier's copy (no decommitment: `SNARK::prove` refuses it)
    pub fn from_bytes(bytes: &[u8]) -> Option<ComputationCommitment> {
        let mut h = ptr::null_mut();
        if unsafe { otti_comp_comm_from_bytes(bytes.as_ptr(), bytes.len(), &mut h) } == 0 { Some(ComputationCommitment { h }) } else { None }
    }
}

pub struct SNARK { pub bytes: Vec<u8> }
impl SNARK {
    /// once per circuit, on the GPU
    pub fn encode(inst: &Instance, gens: &SNARKGens) -> (ComputationCommitment, ComputationDecommitment) {
        let mut h = ptr::null_mut();
        let rc = unsafe { otti_snark_encode(inst.h, gens.h, &mut h) };
        assert_eq!(rc, 0, "otti_snark_encode: {}", last_error());
        (ComputationCommitment { h }, ComputationDecommitment { _private: () })
    }
    pub fn prove(inst: &Instance, comm: &ComputationCommitment, _decomm: &ComputationDecommitment, vars: VarsAssignment, inputs: &InputsAssignment,
                 gens: &SNARKGens, transcript_label: &'static [u8]) -> SNARK {
        let (mut p, mut n) = (ptr::null_mut::<u8>(), 0usize);
        let rc = unsafe {
            otti_snark_prove(inst.h, comm.h, vars.bytes.as_ptr() as *const u8, vars.bytes.len(), inputs.bytes.as_ptr() as *const u8, inputs.bytes.len(), gens.h,
                             transcript_label.as_ptr(), transcript_label.len(), ptr::null(), OTTI_FLAG_GPU, &mut p, &mut n, ptr::null_mut())
        };
        assert_eq!(rc, 0, "otti_snark_prove: {}", last_error());
        let bytes = unsafe { std::slice::from_raw_parts(p, n) }.to_vec();
        unsafe { otti_buf_free(p as *mut c_void) };
        SNARK { bytes }
    }
    pub fn verify(&self, comm: &ComputationCommitment, inputs: &InputsAssignment, transcript_label: &'static [u8], gens: &SNARKGens) -> Result<(), ProofVerifyError> {
        let rc = unsafe {
            otti_snark_verify(comm.h, inputs.bytes.as_ptr() as *const u8, inputs.bytes.len(), gens.h, transcript_label.as_ptr(), transcript_label.len(),
                              self.bytes.as_ptr(), self.bytes.len())
        };
        match rc {
            0 => Ok(()),
            -11 => Err(ProofVerifyError::DecompressionError([0u8; 32])),
            _ => Err(ProofVerifyError::InternalError),
        }
    }
}
