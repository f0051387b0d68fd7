//! Reference-side golden dump for imagepipe_amd (the rows no reference test pins).
//!
//! This is a UNIT TEST FOR THE REFERENCE CRATE (pedrocr/imagepipe 0.5.0), written for the imagepipe_amd repository; it contains no reference code.
//! It feeds the inputs committed under `tests/golden/pin/` through the reference's own `OpDemosaic::run`, `demosaic::full`, `scaling::scaled_demosaic`
//! and rawloader's `CFA::new` / `color_at`, and writes what they produce to `tests/golden/ref/`, where `tests/test_golden.py::test_reference_dump_*`
//! compares the CPU oracle and the HIP kernels with it bit for bit.  It has to live inside the crate because `OpBuffer` and `mod scaling` are not
//! exported (src/lib.rs: `mod buffer; mod scaling;`).
//!
//! Procedure ("Pinning the unpinned rows"):
//!   cp <imagepipe_amd>/bindings/rust/dump_goldens.rs <imagepipe>/src/ipk_dump_goldens.rs
//!   echo '#[cfg(test)] mod ipk_dump_goldens;' >> <imagepipe>/src/lib.rs
//!   IPK_GOLDEN_DIR=<imagepipe_amd>/tests/golden cargo test ipk_dump -- --nocapture
//!   cd <imagepipe_amd> && python -m pytest tests/test_golden.py -k reference_dump          (CPU oracle; add `-m gpu` on an MI355X for the kernels)
//!
//! Files written per case `<name>` of pin/cases.txt (raw little-endian, row-major):
//!   ref/<name>.demosaic.f32   OpDemosaic::run on <name>.mosaic.f32 with settings.demosaic_width/height from the case line (4 channels)
//!   ref/<name>.demosaic.dims  "width height colors" of that buffer
//!   ref/<name>.full.f32       demosaic::full on the same mosaic (4 channels, frame size)
//!   ref/<name>.scaled.f32     scaling::scaled_demosaic to the case's demosaic size (only when it is smaller than the frame)
//!   ref/<name>.cfa48.u8       CFA::new(cfa).color_at(row, col) for row, col in 0..48 (what demosaic.rs:77-90 and scaling.rs:110 index)
//!   ref/<name>.cfa.dims       "width height" of the CFA as rawloader parsed the string
//!   ref/pointwise.*.f32/u8/u16  camera_to_lab, xyz_to_lab, lab_to_xyz, lab_to_rgb, apply / expand_srgb_gamma, two splines, output8bit / 16bit on pin/pointwise.*
//! and, only with `RUSTFLAGS="--cfg ipk_rawimage"` (needs rawloader 0.37's `RawImage` field list, see `raw_image` below):
//!   ref/<name>.gofloat.f32    OpGoFloat::run on <name>.raw.u16 (1 channel)

use crate::opbasics::*;
use std::convert::TryInto;
use std::fs;
use std::path::{Path, PathBuf};

fn golden_dir() -> PathBuf {
  PathBuf::from(std::env::var("IPK_GOLDEN_DIR").expect("set IPK_GOLDEN_DIR to <imagepipe_amd>/tests/golden"))
}

fn read_f32(path: &Path) -> Vec<f32> {
  let bytes = fs::read(path).unwrap_or_else(|e| panic!("{}: {}", path.display(), e));
  bytes.chunks_exact(4).map(|b| f32::from_le_bytes(b.try_into().unwrap())).collect()
}

#[allow(dead_code)]
fn read_u16(path: &Path) -> Vec<u16> {
  let bytes = fs::read(path).unwrap_or_else(|e| panic!("{}: {}", path.display(), e));
  bytes.chunks_exact(2).map(|b| u16::from_le_bytes(b.try_into().unwrap())).collect()
}

fn write_f32(path: &Path, data: &[f32]) {
  let mut bytes = Vec::with_capacity(data.len() * 4);
  for v in data {
    bytes.extend_from_slice(&v.to_le_bytes());
  }
  fs::write(path, bytes).unwrap_or_else(|e| panic!("{}: {}", path.display(), e));
}

struct Case {
  name: String,
  cfa: String,
  width: usize,
  height: usize,
  black: u16,
  white: u16,
  demosaic_width: usize,
  demosaic_height: usize,
}

fn cases(dir: &Path) -> Vec<Case> {
  let text = fs::read_to_string(dir.join("pin/cases.txt")).expect("tests/golden/pin/cases.txt");
  text.lines().filter(|l| !l.trim().is_empty() && !l.starts_with('#')).map(|l| {
    let f: Vec<&str> = l.split_whitespace().collect();
    assert_eq!(f.len(), 8, "malformed case line: {}", l);
    Case {
      name: f[0].to_string(),
      cfa: f[1].to_string(),
      width: f[2].parse().unwrap(),
      height: f[3].parse().unwrap(),
      black: f[4].parse().unwrap(),
      white: f[5].parse().unwrap(),
      demosaic_width: f[6].parse().unwrap(),
      demosaic_height: f[7].parse().unwrap(),
    }
  }).collect()
}

fn mosaic(dir: &Path, c: &Case) -> OpBuffer {
  let data = read_f32(&dir.join(format!("pin/{}.mosaic.f32", c.name)));
  assert_eq!(data.len(), c.width * c.height);
  OpBuffer { width: c.width, height: c.height, colors: 1, monochrome: false, data }
}

#[test]
fn ipk_dump_demosaic_goldens() {
  let dir = golden_dir();
  let out = dir.join("ref");
  fs::create_dir_all(&out).unwrap();
  for c in cases(&dir) {
    let _ = (c.black, c.white);
    // rawloader's reading of the pattern string, as the demosaic and the scaled demosaic see it
    let cfa = CFA::new(&c.cfa);
    let mut table = Vec::with_capacity(48 * 48);
    for row in 0..48 {
      for col in 0..48 {
        table.push(cfa.color_at(row, col) as u8);
      }
    }
    fs::write(out.join(format!("{}.cfa48.u8", c.name)), &table).unwrap();
    fs::write(out.join(format!("{}.cfa.dims", c.name)), format!("{} {}\n", cfa.width, cfa.height)).unwrap();

    // OpDemosaic::run with the case's demosaic size: whichever of its four branches the reference takes
    let op = crate::ops::demosaic::OpDemosaic { cfa: c.cfa.clone() };
    let mut globals = PipelineGlobals::mock(c.width as u32, c.height as u32);
    globals.settings.demosaic_width = c.demosaic_width;
    globals.settings.demosaic_height = c.demosaic_height;
    let res = op.run(&globals, Arc::new(mosaic(&dir, &c)));
    write_f32(&out.join(format!("{}.demosaic.f32", c.name)), &res.data);
    fs::write(out.join(format!("{}.demosaic.dims", c.name)), format!("{} {} {}\n", res.width, res.height, res.colors)).unwrap();

    // the two functions directly
    let full = crate::ops::demosaic::full(CFA::new(&c.cfa), &mosaic(&dir, &c));
    write_f32(&out.join(format!("{}.full.f32", c.name)), &full.data);
    if c.demosaic_width < c.width || c.demosaic_height < c.height {
      let scaled = crate::scaling::scaled_demosaic(CFA::new(&c.cfa), &mosaic(&dir, &c), c.demosaic_width, c.demosaic_height);
      write_f32(&out.join(format!("{}.scaled.f32", c.name)), &scaled.data);
    }
    println!("ipk_dump: {} -> {}x{}x{}", c.name, res.width, res.height, res.colors);
  }
}

// The point-wise functions: the reference's own tests pin them through ROUND TRIPS (gamma and Lab there and back), which a table of another length or
// another interpolation would pass as well.  Here their one-way outputs are written for 4096 pixels: camera_to_lab, xyz_to_lab, lab_to_xyz, lab_to_rgb,
// apply / expand_srgb_gamma on the clamped channel (what OpGamma and run_other call), SplineFunc::interpolate for two curves, output8bit / output16bit.
// (XYZ_LAB_TRANSFORM calls the platform's cbrtf for ratios above 1: a libm other than glibc 2.35 may differ there in the last place -- the Python
// side reports such pixels separately.)
#[test]
fn ipk_dump_pointwise_goldens() {
  let dir = golden_dir();
  let out = dir.join("ref");
  fs::create_dir_all(&out).unwrap();
  let px = read_f32(&dir.join("pin/pointwise.rgbe.f32"));
  let par = read_f32(&dir.join("pin/pointwise.params.f32"));
  let cur = read_f32(&dir.join("pin/pointwise.curves.f32"));
  assert_eq!(px.len() % 4, 0);
  assert_eq!(par.len(), 16);
  assert_eq!(cur.len(), 10);
  let mul = [par[0], par[1], par[2], par[3]];
  let cm = [[par[4], par[5], par[6], par[7]], [par[8], par[9], par[10], par[11]], [par[12], par[13], par[14], par[15]]];
  let n = px.len() / 4;
  let (mut lab, mut xyzlab, mut labxyz, mut rgb, mut gam, mut exp) = (vec![], vec![], vec![], vec![], vec![], vec![]);
  let (mut o8, mut o16) = (Vec::<u8>::new(), Vec::<u8>::new());
  for i in 0..n {
    let p = &px[4 * i..4 * i + 4];
    let (l, a, b) = camera_to_lab(mul, cm, p);
    lab.extend_from_slice(&[l, a, b]);
    let (l2, a2, b2) = xyz_to_lab(p[0], p[1], p[2]);
    xyzlab.extend_from_slice(&[l2, a2, b2]);
    let (x, y, z) = lab_to_xyz(l2, a2, b2);
    labxyz.extend_from_slice(&[x, y, z]);
    let (r, g, bb) = lab_to_rgb(*XYZ_D65_33, &[l, a, b]);
    rgb.extend_from_slice(&[r, g, bb]);
    for c in 0..3 {
      let v = p[c].max(0.0).min(1.0);
      gam.push(apply_srgb_gamma(v));
      exp.push(expand_srgb_gamma(v));
      o8.push(output8bit(p[c]));
      o16.extend_from_slice(&output16bit(p[c]).to_le_bytes());
    }
  }
  write_f32(&out.join("pointwise.camera_to_lab.f32"), &lab);
  write_f32(&out.join("pointwise.xyz_to_lab.f32"), &xyzlab);
  write_f32(&out.join("pointwise.lab_to_xyz.f32"), &labxyz);
  write_f32(&out.join("pointwise.lab_to_rgb.f32"), &rgb);
  write_f32(&out.join("pointwise.apply_srgb_gamma.f32"), &gam);
  write_f32(&out.join("pointwise.expand_srgb_gamma.f32"), &exp);
  fs::write(out.join("pointwise.output8bit.u8"), &o8).unwrap();
  fs::write(out.join("pointwise.output16bit.u16"), &o16).unwrap();
  // curves: one user point, then four (SplineFunc::new adds the ends), evaluated on every input channel value
  let c3 = crate::ops::curves::SplineFunc::new(&[(cur[0], cur[1])]);
  let c6 = crate::ops::curves::SplineFunc::new(&[(cur[2], cur[3]), (cur[4], cur[5]), (cur[6], cur[7]), (cur[8], cur[9])]);
  let (mut s3, mut s6) = (vec![], vec![]);
  for i in 0..n {
    for c in 0..3 {
      s3.push(c3.interpolate(px[4 * i + c]));
      s6.push(c6.interpolate(px[4 * i + c]));
    }
  }
  write_f32(&out.join("pointwise.spline3.f32"), &s3);
  write_f32(&out.join("pointwise.spline6.f32"), &s6);
  println!("ipk_dump: pointwise, {} pixels", n);
}

// OpGoFloat::run_raw is private and reads a rawloader::RawImage.  The literal below lists RawImage's fields as of rawloader 0.37 (decoders/image.rs); if
// the struct has changed, the compiler names the field to fix -- nothing else in this file depends on it, which is why it sits behind its own cfg.
#[cfg(ipk_rawimage)]
fn raw_image(c: &Case, data: Vec<u16>) -> RawImage {
  RawImage {
    make: "ipk".to_string(),
    model: "synthetic".to_string(),
    clean_make: "ipk".to_string(),
    clean_model: "synthetic".to_string(),
    width: c.width,
    height: c.height,
    cpp: 1,
    wb_coeffs: [2.0, 1.0, 1.5, std::f32::NAN],
    whitelevels: [c.white; 4],
    blacklevels: [c.black; 4],
    xyz_to_cam: [[0.0; 3]; 4],
    cfa: CFA::new(&c.cfa),
    crops: [0, 0, 0, 0],
    blackareas: Vec::new(),
    orientation: Orientation::Normal,
    data: RawImageData::Integer(data),
  }
}

#[cfg(ipk_rawimage)]
#[test]
fn ipk_dump_gofloat_goldens() {
  let dir = golden_dir();
  let out = dir.join("ref");
  fs::create_dir_all(&out).unwrap();
  for c in cases(&dir) {
    let data = read_u16(&dir.join(format!("pin/{}.raw.u16", c.name)));
    assert_eq!(data.len(), c.width * c.height);
    let source = ImageSource::Raw(raw_image(&c, data));
    let op = crate::ops::gofloat::OpGoFloat::new(&source);
    let mut globals = PipelineGlobals::mock(c.width as u32, c.height as u32);
    globals.image = source;
    let res = op.run(&globals, Arc::new(OpBuffer::default()));
    assert_eq!((res.width, res.height, res.colors), (c.width, c.height, 1));
    write_f32(&out.join(format!("{}.gofloat.f32", c.name)), &res.data);
  }
}
