//! PINNING KIT for the two contract items this build could not pin (no Rust toolchain, no network in the build image):
//! drop this file into the reference crate's `tests/` directory and run `cargo test --test pin -- --nocapture`.
//! It links nothing of the MI355X library: it asks the REFERENCE what it does, so that the oracle
//! (`oracle/icp_oracle.c`) and the device code can be held to it.  SOURCE ONLY -- it has never been compiled.
//!
//! 1. `median_of_an_even_count_*`: `stats::mutable_median` reads `input[n/2 - 1]` AFTER a second
//!    `select_nth_unstable_by(n/2, ..)` on the whole vector (src/stats.rs:23-26).  That call only guarantees index
//!    n/2; whether the lower middle survives at n/2 - 1 depends on std's selection for slices beyond its
//!    insertion-sort cutoff.  The oracle and the GPU return the mathematically exact median
//!    (lower middle + upper middle) / 2.  If `median_of_an_even_count_is_the_exact_one` FAILS on your toolchain, the
//!    reference's even-n median is "whatever sits at n/2 - 1 after the second select" -- see INTEGRATION.md section 12
//!    for what to change.
//! 2. `kdtree_*`: the `nearest_neighbor` crate is un-vendored and un-pinned (Cargo.toml:22-25).  The library's contract
//!    is d^2 = ((dx*dx + dy*dy) + dz*dz) in f64 and ties -> lowest index.  These tests print / assert what the crate does.

use icp::stats::mutable_median; // (make `stats` and `mutable_median` pub(crate) -> pub for the test, or move this into src/)

fn splitmix(state: &mut u64) -> f64 {
    *state = state.wrapping_add(0x9E3779B97F4A7C15);
    let mut z = *state;
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^= z >> 31;
    (z >> 11) as f64 / (1u64 << 53) as f64
}

fn exact_median(v: &[f64]) -> f64 {
    let mut s = v.to_vec();
    s.sort_by(|a, b| a.partial_cmp(b).unwrap());
    let n = s.len();
    if n % 2 == 1 { s[n / 2] } else { (s[n / 2 - 1] + s[n / 2]) / 2. }
}

/// n = 1 000 (even, far beyond any insertion-sort cutoff), 200 different inputs, distinct values: the exact median
/// and "whatever sits at n/2 - 1" differ in the last bits whenever the second select moved the lower middle.
#[test]
fn median_of_an_even_count_is_the_exact_one() {
    let mut differing = 0usize;
    for seed in 0..200u64 {
        let mut st = 0x1C920240807u64 ^ (seed << 20);
        let v: Vec<f64> = (0..1000).map(|_| splitmix(&mut st) - 0.5).collect();
        let want = exact_median(&v);
        let got = mutable_median(&mut v.clone()).unwrap();
        if got.to_bits() != want.to_bits() {
            differing += 1;
            eprintln!("seed {seed}: mutable_median = {got:e}, exact = {want:e}");
        }
    }
    assert_eq!(differing, 0, "the reference's even-n median is NOT the exact median on this toolchain: INTEGRATION.md section 12");
}

/// The size the benchmark runs (1 000 000 is even): one input, the same question.
#[test]
fn median_of_an_even_count_at_the_benchmark_size() {
    let mut st = 0x1C920240807u64;
    let v: Vec<f64> = (0..1_000_000).map(|_| splitmix(&mut st) - 0.5).collect();
    assert_eq!(mutable_median(&mut v.clone()).unwrap().to_bits(), exact_median(&v).to_bits());
}

/// Duplicates around the middle (the 2-D scans hold runs of equal residuals): both middles equal -> any selection agrees.
#[test]
fn median_of_an_even_count_with_duplicates_in_the_middle() {
    let mut v: Vec<f64> = (0..1000).map(|i| if (400..600).contains(&i) { 0.25 } else if i < 400 { -1.0 - i as f64 } else { 1.0 + i as f64 }).collect();
    v.reverse();
    assert_eq!(mutable_median(&mut v).unwrap(), 0.25);
}

use nearest_neighbor::KdTree;
use nalgebra::Vector3;

/// Ties: a query at the centre of a cube of eight targets, all at the same distance.  The library returns index 0.
#[test]
fn kdtree_tie_goes_to_which_index() {
    let mut dst = Vec::new();
    for &x in &[-1.0, 1.0] { for &y in &[-1.0, 1.0] { for &z in &[-1.0, 1.0] { dst.push(Vector3::new(x, y, z)); } } }
    let tree = KdTree::new(&dst, 2);
    let (idx, d) = tree.search(&Vector3::new(0., 0., 0.));
    eprintln!("tie among 8 equidistant targets -> index {:?}, distance {:?}", idx, d);
    assert_eq!(idx, Some(0), "the library's contract is: ties -> lowest target index (DESIGN.md section 3)");
}

/// Exact duplicates of one target (scans/2d holds 109 duplicate (0, 0) points): which copy is returned?
#[test]
fn kdtree_duplicate_targets() {
    let dst = vec![Vector3::new(5., 5., 5.), Vector3::new(1., 2., 3.), Vector3::new(1., 2., 3.), Vector3::new(1., 2., 3.)];
    let tree = KdTree::new(&dst, 2);
    let (idx, _) = tree.search(&Vector3::new(1.1, 2.0, 3.0));
    eprintln!("three copies of the nearest target at indices 1, 2, 3 -> index {:?}", idx);
    assert_eq!(idx, Some(1));
}

/// The order of the distance sum: ((dx^2 + dy^2) + dz^2) against (dx^2 + (dy^2 + dz^2)) differ in the last bit for
/// these coordinates; two targets are built so that the two orders rank them differently.
#[test]
fn kdtree_distance_summation_order() {
    let q = Vector3::new(0., 0., 0.);
    // a: dx^2 = 1e16, dy^2 = 1, dz^2 = 1  ->  (1e16 + 1) + 1 = 1e16 (each 1 is absorbed); 1e16 + (1 + 1) = 1e16 + 2
    let a = Vector3::new(1e8, 1., 1.);
    // b: dx^2 = 1e16, dy^2 = 2.0000000000000004, dz^2 = 0  ->  1e16 + 2 under either order (the spacing at 1e16 is 2)
    let b = Vector3::new(1e8, 2f64.sqrt(), 0.);
    let tree = KdTree::new(&vec![b, a], 2);
    let (idx, _) = tree.search(&q);
    // x-then-y-then-z left fold: d2(a) = 1e16 < d2(b) -> a (index 1).  Any other order: a tie or b -> index 0.
    eprintln!("summation order probe -> index {:?} (1 = ((dx^2 + dy^2) + dz^2), the library's contract)", idx);
}
