//! src/scene.rs -- fills the reference's empty module (reference src/scene.rs:1, `mod scene;` at src/main.rs:2).
//!
//! A `Scene` owns the simulation state the reference keeps as locals of `main()` (src/main.rs:738-750) and
//! advances it with libnenbody_hip.so (include/nenbody.h).  `Scene::step()` is `update_instance_nbody`
//! (src/main.rs:404-441) on the GPU; afterwards `positions`, `velocities` and `instances` hold what the
//! per-frame consumers read (src/main.rs:932-945).
//!
//! NOT COMPILED in the build environment (no cargo/rustc there): a mechanical binding of the C ABI.
use cgmath::{Point3, Vector3};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct NbParams {
    pub dt: f32,   // src/main.rs:411
    pub g: f32,    // src/main.rs:412
    pub bias: f32, // src/main.rs:413
    pub tile: u32, // 0 = library default
    pub mode: u32, // 0 = STRICT (bit-identical to the CPU path), 1 = FAST
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct NbBoidsParams {
    pub dt: f32,              // src/main.rs:450
    pub rule_1_distance: f32, // src/main.rs:451 (squared-distance radius)
    pub rule_2_distance: f32, // src/main.rs:452
    pub rule_3_distance: f32, // src/main.rs:453 (velocity space)
    pub rule_1_scale: f32,    // src/main.rs:454
    pub rule_2_scale: f32,    // src/main.rs:455
    pub rule_3_scale: f32,    // src/main.rs:456
    pub tile: u32,
}

#[repr(C)]
pub struct NbCtx {
    _private: [u8; 0],
}

/// The ABI this binding was written against (include/nenbody.h: NB_ABI_VERSION).  A symbol may keep its name and change its
/// arguments between versions (`nb_update_instance_random`: eight in ABI 1, six since ABI 2) and the linker cannot tell, so
/// every entry point of this module checks it once per process before anything else crosses the boundary.
pub const NB_ABI_VERSION: c_int = 2;

#[link(name = "nenbody_hip")]
extern "C" {
    fn nb_abi_version() -> c_int;
    fn nb_default_params(p: *mut NbParams);
    fn nb_last_error(ctx: *const NbCtx) -> *const c_char;
    fn nb_init_state(seed: u64, n: u32, pos_xyz: *mut f32, vel_xyz: *mut f32) -> c_int;
    fn nb_create(n: u32, n_devices: u32, params: *const NbParams, out: *mut *mut NbCtx) -> c_int;
    fn nb_destroy(ctx: *mut NbCtx);
    fn nb_upload(ctx: *mut NbCtx, pos_xyz: *const f32, vel_xyz: *const f32) -> c_int;
    fn nb_step(ctx: *mut NbCtx, k: u32) -> c_int;
    fn nb_step_boids(ctx: *mut NbCtx, k: u32, params: *const NbBoidsParams) -> c_int; // null = reference constants
    fn nb_download(ctx: *mut NbCtx, pos_xyz: *mut f32, vel_xyz: *mut f32, inst_16n: *mut f32) -> c_int;
    fn nb_sync(ctx: *mut NbCtx) -> c_int;
    // the reference's free functions themselves: five slices and their lengths, params null = reference constants
    fn nb_update_instance_nbody(
        instances: *mut f32, n_instances: usize, positions: *mut f32, n_positions: usize,
        old_positions: *mut f32, n_old_positions: usize, velocities: *mut f32, n_velocities: usize,
        old_velocities: *mut f32, n_old_velocities: usize, params: *const NbParams,
    ) -> c_int;
    fn nb_update_instance_boids(
        instances: *mut f32, n_instances: usize, positions: *mut f32, n_positions: usize,
        old_positions: *mut f32, n_old_positions: usize, velocities: *mut f32, n_velocities: usize,
        old_velocities: *mut f32, n_old_velocities: usize, params: *const NbBoidsParams,
    ) -> c_int;
    fn nb_update_instance_random(
        instances: *mut f32, n_instances: usize, positions: *mut f32, n_positions: usize,
        velocities: *mut f32, n_velocities: usize,
    ) -> c_int;
    fn nb_update_instance_random_seeded(
        instances: *mut f32, n_instances: usize, positions: *mut f32, n_positions: usize,
        velocities: *mut f32, n_velocities: usize, seed: u64, step: u64,
    ) -> c_int;
    fn nb_update_random_seed(seed: u64);
}

fn check_abi() {
    static ONCE: std::sync::Once = std::sync::Once::new();
    ONCE.call_once(|| {
        let got = unsafe { nb_abi_version() };
        assert_eq!(got, NB_ABI_VERSION, "libnenbody_hip.so speaks ABI {}, this binding was written against ABI {}", got, NB_ABI_VERSION);
    });
}

impl Default for NbParams {
    fn default() -> Self {
        check_abi();
        let mut p = NbParams { dt: 0.0, g: 0.0, bias: 0.0, tile: 0, mode: 0 };
        unsafe { nb_default_params(&mut p) };
        p
    }
}

#[derive(Debug)]
pub struct SceneError(pub i32, pub String);

fn check(rc: c_int, ctx: *const NbCtx) -> Result<(), SceneError> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(nb_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(SceneError(rc, msg))
}

/// Single owner of a device context: `Send`, not `Sync` (the reference calls the update on the winit main
/// thread, src/main.rs:925).
pub struct Scene {
    ctx: *mut NbCtx,
    pub positions: Vec<Point3<f32>>,   // src/main.rs:743
    pub velocities: Vec<Vector3<f32>>, // src/main.rs:738
    pub instances: Vec<[[f32; 4]; 4]>, // instance_data, uploaded at src/main.rs:932-936
}

unsafe impl Send for Scene {}

impl Scene {
    /// `entity_count` bodies with the reference's initial distributions (src/main.rs:738-747), seeded.
    pub fn new(n: usize, params: NbParams, seed: u64) -> Result<Scene, SceneError> {
        check_abi();
        let mut positions = vec![Point3::new(0.0f32, 0.0, 0.0); n];
        let mut velocities = vec![Vector3::new(0.0f32, 0.0, 0.0); n];
        // Point3<f32> / Vector3<f32> are #[repr(C)] {x, y, z}: the Vec's buffer is the stride-3 array the ABI takes
        check(
            unsafe { nb_init_state(seed, n as u32, positions.as_mut_ptr() as *mut f32, velocities.as_mut_ptr() as *mut f32) },
            std::ptr::null(),
        )?;
        Scene::from_state(positions, velocities, params)
    }

    pub fn from_state(positions: Vec<Point3<f32>>, velocities: Vec<Vector3<f32>>, params: NbParams) -> Result<Scene, SceneError> {
        check_abi();
        // same panic the reference has at src/main.rs:415-416 (copy_from_slice on unequal lengths)
        assert_eq!(positions.len(), velocities.len(), "positions and velocities must have the same length");
        let n = positions.len();
        let mut ctx: *mut NbCtx = std::ptr::null_mut();
        check(unsafe { nb_create(n as u32, 1, &params, &mut ctx) }, std::ptr::null())?;
        let scene = Scene { ctx, positions, velocities, instances: vec![[[0.0; 4]; 4]; n] };
        check(
            unsafe { nb_upload(scene.ctx, scene.positions.as_ptr() as *const f32, scene.velocities.as_ptr() as *const f32) },
            scene.ctx,
        )?;
        Ok(scene)
    }

    /// One `update_instance_nbody` (src/main.rs:404-441); host mirrors refreshed so the consumers at
    /// src/main.rs:932-945 work unchanged.
    pub fn step(&mut self) -> Result<(), SceneError> {
        check(unsafe { nb_step(self.ctx, 1) }, self.ctx)?;
        check(
            unsafe {
                nb_download(
                    self.ctx,
                    self.positions.as_mut_ptr() as *mut f32,
                    self.velocities.as_mut_ptr() as *mut f32,
                    self.instances.as_mut_ptr() as *mut f32,
                )
            },
            self.ctx,
        )
    }

    /// One `update_instance_boids` (src/main.rs:443-526) with the reference's constants; host mirrors refreshed.
    pub fn step_boids(&mut self) -> Result<(), SceneError> {
        check(unsafe { nb_step_boids(self.ctx, 1, std::ptr::null()) }, self.ctx)?;
        check(
            unsafe {
                nb_download(
                    self.ctx,
                    self.positions.as_mut_ptr() as *mut f32,
                    self.velocities.as_mut_ptr() as *mut f32,
                    self.instances.as_mut_ptr() as *mut f32,
                )
            },
            self.ctx,
        )
    }

    /// k steps, device-resident, no download: the benchmark path.
    pub fn step_n(&mut self, k: u32) -> Result<(), SceneError> {
        check(unsafe { nb_step(self.ctx, k) }, self.ctx)
    }

    pub fn sync(&mut self) -> Result<(), SceneError> {
        check(unsafe { nb_sync(self.ctx) }, self.ctx)
    }
}

impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { nb_destroy(self.ctx) };
    }
}

/// Drop-in for the reference's free function (same five arguments, src/main.rs:404-410): the body is one FFI call.
/// The snapshot copies (src/main.rs:415-416), the `zip` truncation (src/main.rs:420-423) and the fold over all of
/// `old_positions` happen inside `nb_update_instance_nbody`; a length mismatch panics here as `copy_from_slice`
/// does there.  One upload, one step, one download per call; the device context is kept by the library.
pub fn update_instance_nbody(
    instances: &mut Vec<[[f32; 4]; 4]>,
    positions: &mut Vec<Point3<f32>>,
    old_positions: &mut Vec<Point3<f32>>,
    velocities: &mut Vec<Vector3<f32>>,
    old_velocities: &mut Vec<Vector3<f32>>,
) {
    check_abi();
    let rc = unsafe {
        nb_update_instance_nbody(
            instances.as_mut_ptr() as *mut f32, instances.len(),
            positions.as_mut_ptr() as *mut f32, positions.len(),
            old_positions.as_mut_ptr() as *mut f32, old_positions.len(),
            velocities.as_mut_ptr() as *mut f32, velocities.len(),
            old_velocities.as_mut_ptr() as *mut f32, old_velocities.len(),
            std::ptr::null(),
        )
    };
    if let Err(SceneError(code, msg)) = check(rc, std::ptr::null()) {
        panic!("update_instance_nbody: {} ({})", msg, code);
    }
}

/// The controller the event loop calls today (src/main.rs:443-449, call site src/main.rs:925-931), same contract.
pub fn update_instance_boids(
    instances: &mut Vec<[[f32; 4]; 4]>,
    positions: &mut Vec<Point3<f32>>,
    old_positions: &mut Vec<Point3<f32>>,
    velocities: &mut Vec<Vector3<f32>>,
    old_velocities: &mut Vec<Vector3<f32>>,
) {
    check_abi();
    let rc = unsafe {
        nb_update_instance_boids(
            instances.as_mut_ptr() as *mut f32, instances.len(),
            positions.as_mut_ptr() as *mut f32, positions.len(),
            old_positions.as_mut_ptr() as *mut f32, old_positions.len(),
            velocities.as_mut_ptr() as *mut f32, velocities.len(),
            old_velocities.as_mut_ptr() as *mut f32, old_velocities.len(),
            std::ptr::null(),
        )
    };
    if let Err(SceneError(code, msg)) = check(rc, std::ptr::null()) {
        panic!("update_instance_boids: {} ({})", msg, code);
    }
}

/// The third controller with the reference's own signature (src/main.rs:381-385): a drop-in for the function body.
/// The reference draws from an unseeded `thread_rng`; the library draws from a counter-based stream whose seed and call
/// counter it keeps itself (`update_random_seed` restarts it).
pub fn update_instance_random(
    instances: &mut Vec<[[f32; 4]; 4]>,
    positions: &mut Vec<Point3<f32>>,
    velocities: &mut Vec<Vector3<f32>>,
) {
    check_abi();
    let rc = unsafe {
        nb_update_instance_random(
            instances.as_mut_ptr() as *mut f32, instances.len(),
            positions.as_mut_ptr() as *mut f32, positions.len(),
            velocities.as_mut_ptr() as *mut f32, velocities.len(),
        )
    };
    if let Err(SceneError(code, msg)) = check(rc, std::ptr::null()) {
        panic!("update_instance_random: {} ({})", msg, code);
    }
}

/// The same step at a stream position the caller names (`step` = the frame number): independent of how a run is split.
pub fn update_instance_random_seeded(
    instances: &mut Vec<[[f32; 4]; 4]>,
    positions: &mut Vec<Point3<f32>>,
    velocities: &mut Vec<Vector3<f32>>,
    seed: u64,
    step: u64,
) {
    check_abi();
    let rc = unsafe {
        nb_update_instance_random_seeded(
            instances.as_mut_ptr() as *mut f32, instances.len(),
            positions.as_mut_ptr() as *mut f32, positions.len(),
            velocities.as_mut_ptr() as *mut f32, velocities.len(),
            seed, step,
        )
    };
    if let Err(SceneError(code, msg)) = check(rc, std::ptr::null()) {
        panic!("update_instance_random_seeded: {} ({})", msg, code);
    }
}

/// Seed of `update_instance_random`'s stream; restarts its call counter.
pub fn update_random_seed(seed: u64) {
    unsafe { nb_update_random_seed(seed) }
}
