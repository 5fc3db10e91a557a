/*
 * aerial_gym_hip.h -- C ABI of libaerialgym_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary of the hot path.  The reference
 * (ntnu-arl/aerial_gym_simulator) has no FFI of its own: its "plugin API" is
 * Python (string registries + manager classes + one dict of aliasing tensors,
 * SURVEY.md section 8b).  Behind those Python classes the reference calls three
 * device back-ends -- PyTorch op chains, NVIDIA Warp kernels, and the Isaac Gym
 * (PhysX) C API.  Every entry point below replaces one such call site and cites
 * it.  The Python host (aerial_gym_simulator_amd/) binds them with ctypes, see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, a hipStream_t passed as void*;
 *     no torch / C++ types cross the boundary.
 *   - return 0 on success, negative on error (AGX_E_*); text via agx_last_error().
 *   - no allocation, no ownership transfer, stream-ordered, no host sync.
 *   - fp32 everywhere; quaternions are xyzw like the reference.
 *   - env state is SoA, component-major:  X[c * num_envs + env]
 *     (the reference is AoS [N, C]; the Python host exposes transposed views so
 *      `robot_position[N,3]` etc. keep their reference shapes).
 *   - images are [N, S, H, W] row-major exactly like the reference
 *     (warp_cam.py:134-149 writes pixels[env, cam, y, x]).
 */
#ifndef AERIAL_GYM_HIP_H
#define AERIAL_GYM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGX_ABI_VERSION 12
#define AGX_MAX_MOTORS 8
#define AGX_MAX_ACTIONS 8
#define AGX_MAX_SUBSTEPS 32
#define AGX_MAX_BODIES 24   /* rigid bodies (links) of one robot: base_quadrotor 9, base_octarotor 17 */

enum {
  AGX_OK = 0,
  AGX_E_ARG = -1,     /* bad argument (null pointer, size out of range)  */
  AGX_E_LAUNCH = -2,  /* hipLaunch / runtime error                       */
  AGX_E_UNSUPPORTED = -3
};

/* controller ids, one per class registered in aerial_gym/control/__init__.py:42-100 */
enum {
  AGX_CTRL_NONE = 0,          /* no_control                                            */
  AGX_CTRL_POSITION = 1,      /* LeePositionController     position_control.py:20      */
  AGX_CTRL_VELOCITY = 2,      /* LeeVelocityController     velocity_control.py:18      */
  AGX_CTRL_ATTITUDE = 3,      /* LeeAttitudeController     attitude_control.py:16      */
  AGX_CTRL_RATES = 4,         /* LeeRatesController        rates_control.py:16         */
  AGX_CTRL_ACCELERATION = 5,  /* LeeAccelerationController acceleration_control.py:16  */
  AGX_CTRL_VEL_STEERING = 6,  /* LeeVelocitySteeringAngleController  :15               */
  AGX_CTRL_FULLY_ACTUATED = 7,/* FullyActuatedController   fully_actuated_control.py:14 */
  AGX_CTRL_WRENCH = 8         /* EXTERNAL controller: any class registered through controller_registry.register_controller
                                 (registry/controller_registry.py:12-52) whose __call__(action) returns the body wrench
                                 [N, 6] (base_lee_controller.py:92-93).  The host evaluates it (torch) once per physics
                                 sub-step on freshly updated state tensors and hands its OUTPUT to agx_env_step as
                                 `actions_in` [N][6] with k_substeps = 1 and AgxEnvBuffers.launch_flags; allocation, motor
                                 model, drag, disturbance, integration, collision run in the kernel as usual.  num_actions
                                 stays the width of robot_actions (maintained by the host in this mode).               */
};

/* Constants shared by all envs.  Sources: config/robot_config/ *.py,
 * config/controller_config/ *.py, config/sim_config/base_sim_config.py,
 * resources/robots/<robot>/<robot>.urdf (composite mass/inertia computed as in
 * robots/robot_manager.py:295-435).                                          */
typedef struct AgxRobotParams {
  int32_t num_motors;
  int32_t num_actions;
  int32_t controller;
  int32_t root_link_mode;               /* force_application_level == "root_link" */
  float dt;
  float dt_over_6;                      /* float(dt / 6.0) with dt the config's double: the reference forms the RK4 weight
                                           from python scalars, `(dt / 6.0) * (k1 + ...)` (motor_model.py:198) -- not
                                           float(dt) / 6.0f, which is 1 ulp away for dt = 0.01                       */
  float gravity[3];
  float mass;
  float inertia[9];                     /* row-major, body frame, about COM       */
  float inertia_inv[9];
  float alloc[6 * AGX_MAX_MOTORS];      /* allocation_matrix, 6 x M row-major     */
  float alloc_pinv[AGX_MAX_MOTORS * 6]; /* pinv, M x 6 row-major                  */
  float wrench_map[6 * AGX_MAX_MOTORS]; /* body wrench per unit motor thrust when
                                           forces are applied at the motor links  */
  float motor_dir[AGX_MAX_MOTORS];
  float cq;
  int32_t use_rps;
  int32_t use_discrete_approximation;
  int32_t integration_rk4;
  float min_thrust, max_thrust, max_rate;
  float max_yaw_rate;
  float lin_drag_linear[3], lin_drag_quadratic[3];
  float ang_drag_linear[3], ang_drag_quadratic[3];
  float linear_damping, angular_damping;
  float max_linear_velocity, max_angular_velocity;
  float collision_radius;
  /* values used when the corresponding per-env buffer pointer in AgxEnvBuffers is NULL
   * (parameter not randomised: min == max in the config) -- saves the HBM reads          */
  float gains_uniform[12];              /* K_pos K_vel K_rot K_angvel                     */
  float tau_inc_uniform, tau_dec_uniform;
} AgxRobotParams;

/* Per-env device buffers of the dynamics path (all SoA, fp32 unless noted).
 * Keys in parentheses are the reference's global_tensor_dict names.          */
typedef struct AgxEnvBuffers {
  float *state;          /* [13][N] p q v w            (robot_state_tensor)            */
  float *derived;        /* [16][N] euler(3) qveh(4) vveh(3) vbody(3) wbody(3)
                            (robot_euler_angles, robot_vehicle_orientation,
                             robot_vehicle_linvel, robot_body_linvel, robot_body_angvel) */
  float *actions;        /* [A][N] (robot_actions)  -- written from actions_in        */
  float *prev_actions;   /* [A][N] (robot_prev_actions)                               */
  float *motor_thrust;   /* [M][N] MotorModel.current_motor_thrust                    */
  float *motor_kT;       /* [M][N] motor_thrust_constant (use_rps only)               */
  float *motor_tau_inc;  /* [M][N] motor_time_constants_increasing, or NULL (uniform)   */
  float *motor_tau_dec;  /* [M][N] motor_time_constants_decreasing, or NULL (uniform)   */
  float *gains;          /* [12][N] K_pos K_vel K_rot K_angvel (current values), or NULL (uniform) */
  float *wrench_cmd;     /* [6][N]  controller output of the LAST sub-step, or NULL     */
  uint8_t *crashes;      /* [N] bool (crashes)                                        */
  uint8_t *truncations;  /* [N] bool (truncations)                                    */
  int32_t *sim_steps;    /* [N]   EnvManager.sim_steps                                */
  uint8_t *reset_mask;   /* [N]   envs to reset = crashes*reset_on_collision | truncations
                            (env_manager.py:364-371), written by the task reward kernels          */
  int32_t *reset_flag;   /* [2]   device flags, double buffered by env-step parity: the task reward
                            (kernel or fused epilogue) ORs 1 into reset_flag[flag_parity] when any
                            env must reset; agx_reset_masked / agx_post_step_* read that entry and
                            clear the other one for the next step.  No host sync, no memset.     */
  int32_t flag_parity;   /* 0 / 1, toggled by the host once per env step                         */
  int32_t *episode_count;/* [N]   number of resets of each env (device RNG counter), or NULL     */
  float *bounds_min;     /* [3][N] (env_bounds_min)                                    */
  float *bounds_max;     /* [3][N] (env_bounds_max)                                    */
  /* optional inputs */
  const float *disturb;  /* [k][7][N] per sub-step (bernoulli outcome, 6 x U01), or NULL:
                            with disturb_prob > 0 the kernel then draws them itself (Philox,
                            counter = (env, step_counter, sub-step))                        */
  float disturb_max[6];
  float disturb_prob;    /* cfg.disturbance.prob_apply_disturbance, 0 = disabled           */
  int32_t step_counter;  /* env steps taken so far (host increments once per env step)      */
  uint64_t rng_seed;     /* key of the device generator                                     */
  const float *boxes;    /* [K][11][N] obstacle OBBs centre(3) quat(4) half(3) bounding radius(1), or NULL */
  int32_t num_boxes;
  /* optional multi-GPU exchange rows (ONE gather per env step).  When step_rows[p] is
     set, every kernel that writes the task observation (agx_post_step_position, agx_obs_position,
     agx_obs_navigation) also writes row i of step_rows[flag_parity] =
       obs(obs_dim) | reward | terminated (crashes) | truncated     ([N][obs_dim + 3] row-major)
     so the send buffer of the collective costs no extra launch.  Double buffered by step parity so
     the gather of step t may overlap step t+1.                                                 */
  float *step_rows[2];
  const float *step_reward; /* [N] the task's reward buffer (required when step_rows is set)     */
  /* optional, with step_rows: uint32 [4] in device memory, zero-initialised by the host.  The LAST
     wave of a row-writing kernel to finish stores step_signal[flag_parity] = step_counter + 1 (release,
     agent scope) after every row of that launch is visible device-wide ([2] is the kernel's own arrival
     counter).  The exchange's communication stream spins on it (agx_exchange_post, signal != NULL)
     instead of a cross-queue event: no host call per step for the producer side.                   */
  uint32_t *step_signal;
  /* optional PEER PUSH of the exchange rows (agx_exchange_create_push): step_rows[flag_parity] then points at THIS rank's own
     slice of a slot of its receive buffer, and every row store is repeated at the same offset in each peer's buffer
     (address + push_delta[j]: the peers' buffers are mapped into this process through hipIpcMemHandle) -- the rows of all
     ranks meet in every rank's buffer with no collective and no extra launch.  The last workgroup of a row-writing kernel
     then stores push_seq into entry push_flag_index of every rank's flag array (system-scope release).  Before its first
     row store such a kernel waits (device side, bounded) until entries push_wait_index .. + push_world of its OWN flag array
     have reached push_wait_seq: the slot it is about to overwrite has been vacated by every rank (0 = no wait).          */
  int64_t push_delta[7];       /* byte offsets own receive buffer -> peer j's, j < push_world - 1                           */
  uint32_t *push_flags[8];     /* flag arrays of ranks 0 .. push_world - 1 (own included), [slots][world] each             */
  int32_t push_world;          /* 0 = no peer push                                                                        */
  int32_t push_rank;
  int32_t push_flag_index;     /* slot * world + rank of this step                                                        */
  int32_t push_wait_index;     /* wait_slot * world                                                                       */
  int32_t push_pub_index;      /* flag entry of the PREVIOUS step's rows (published by the head of this step's first kernel) */
  uint32_t push_pub_seq;       /* ... and their sequence number (0 = nothing to publish)                                  */
  uint32_t push_seq;           /* sequence number of this step's rows (1, 2, ...)                                         */
  uint32_t push_wait_seq;
  uint32_t *push_timed_out;    /* device-visible host word: a wait that gave up stores the sequence number it waited for  */
  char *push_base;             /* this rank's receive buffer [push_slots][push_world][N][row]; with push_slice_bytes = N * row * 4
                                  and push_slots it lets agx_push_advance (called by agx_position_task_step) move to the next
                                  step without the host touching a field                                                  */
  int64_t push_slice_bytes;
  int32_t push_slots;
  float *body_force;     /* optional [3][N]: net applied (non-gravitational) force of the LAST sub-step in
                            the body frame = allocator output + drag + disturbance; read by agx_imu_update */
  const int32_t *step_counter_dev; /* optional, device memory: when set the kernels read the env-step index from here
                            instead of `step_counter` (whose value is frozen in a captured hipGraph); advance it with
                            agx_step_counter_advance as the last launch of the step                                    */
  int32_t env_index_base;/* global index of env 0 of these buffers (sharded runs: this rank owns global envs
                            [env_index_base, env_index_base + N)): first counter word of the device generator, so that
                            a draw is a function of (seed, GLOBAL env, episode | step, stream) whatever the sharding   */
  int32_t launch_flags;  /* 0 for the fused step.  An env step split over several agx_env_step launches (AGX_CTRL_WRENCH:
                            one per physics sub-step): bit 0 = not the first launch (crash flags accumulate, env_manager.py:
                            426-428), bit 1 = not the last launch (no sim_steps += 1, truncation or task epilogue yet),
                            bits 8-15 = index of the first physics sub-step of this launch (disturbance draws / rows).
                            bit 2 (AGX_LAUNCH_LEAN, fused step of a built-in controller, no navigation reward) = the tensors
                            that exist only to be looked at through the tensor dict are not maintained: derived[0..9]
                            (Euler angles, vehicle quaternion, vehicle-frame velocity), actions, prev_actions -- 88 of the
                            ~330 bytes an env moves per step; agx_update_states recomputes the derived tensors from the
                            current state on demand (the host mirror does that when a dict key is read).  The one-lane
                            kernels implement it (it is meant for batches far above 65 536 envs, where bytes matter).
                            bit 3 (AGX_LAUNCH_BODY_WRENCH, with AGX_CTRL_WRENCH only) = EXTERNAL ROBOT: `actions_in` [N][6] is
                            the NET body-frame wrench on the rigid composite about its centre of mass -- what the robot
                            object's step() left in robot_force_tensor / robot_torque_tensor, reduced by
                            agx_net_body_wrench -- and goes straight to the integrator: no allocation, motor model, drag or
                            disturbance in this launch (the robot's step() did whatever it does about them).             */
} AgxEnvBuffers;
#define AGX_LAUNCH_LEAN 4
#define AGX_LAUNCH_BODY_WRENCH 8

const char *agx_last_error(void);
int agx_abi_version(void);
/* Identity of the BINARY: the hash (sha256[:16]) of the kernel sources, this header and the compile flags the library
 * was built from, embedded at compile time (aerial_gym_simulator_amd/_build.py).  Callers compare it with the hash of
 * the sources they see (`_build.source_hash()`): a stale library shipped next to newer sources is detected whatever its
 * file times say; bench.py stamps committed counter files with it.                                                     */
const char *agx_build_id(void);

/* Process-wide tuning / A-B options of the library (none changes a result: every setting is bit-identical, tests hold that).
 * They replace the environment variables rounds 1-5 read inside the launch paths (AGX_ENV_STEP_QUAD, AGX_RAY_SPLIT): the
 * library reads NO environment variable.
 *   "env_step_quad"  1 (default): the four-lanes-per-env kernels where they apply; 0: one lane per env everywhere
 *   "ray_split"      0 (default): workgroups per (env, sensor) image by the launch policy (csrc/agx_raycast.hip
 *                    ray_split_policy); n > 0: that many (clamped to the image's tiles)
 * agx_set_option returns AGX_E_ARG for an unknown name or a value out of range; agx_get_option reads the current value.  */
int agx_set_option(const char *name, int value);
int agx_get_option(const char *name, int *value);

/* ---- dynamics -------------------------------------------------------------------
 * agx_dynamics_substeps: `k` physics sub-steps of every env, fused in one launch.
 * Replaces, per sub-step: RobotManagerIGE.pre_physics_step (robot_manager.py:486-489),
 * BaseMultirotor.step (base_multirotor.py:296-307: update_states, clip, controller,
 * ControlAllocator.allocate_output control_allocation.py:52-114, MotorModel
 * motor_model.py:88-138, simulate_drag, apply_disturbance),
 * gym.apply_rigid_body_force_tensors + gym.simulate + refresh_* (IGE_env_manager.py:
 * 444-449,477,486-495) and EnvManager.compute_observations (env_manager.py:358-362);
 * plus reset_tensors / sim_steps += 1 of EnvManager.step (env_manager.py:399-432).
 * actions_in: [N][A] row-major, exactly the tensor the policy hands to task.step().  */
int agx_dynamics_substeps(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs,
                          const float *actions_in, int k_substeps, void *stream);

/* Task epilogue fused into the env step (same arithmetic as agx_reward_position /
 * agx_reward_navigation, evaluated on the registers of the dynamics kernel).             */
enum { AGX_TASK_NONE = 0, AGX_TASK_POSITION = 1, AGX_TASK_NAVIGATION = 2 };
typedef struct AgxTaskArgs {
  int32_t kind;               /* AGX_TASK_*                                                */
  int32_t episode_len;        /* truncations = sim_steps > episode_len                     */
  int32_t reset_on_collision; /* cfg.env.reset_on_collision                                */
  float curriculum_progress;  /* navigation only                                           */
  const float *target;        /* [3][N]                                                    */
  float *reward;              /* [N]                                                       */
  float *pos_err;             /* [3][N] navigation only (in/out)                           */
  float *prev_pos_err;        /* [3][N] navigation only (out)                              */
  float rp[18];               /* navigation reward parameters                              */
  /* navigation bookkeeping in the same epilogue (navigation_task.py:311-326; what agx_nav_bookkeeping does as a launch of its
   * own): all three NULL = not done here.  successes = truncated & |target - p| < success_radius & not crashed; timeouts =
   * truncated & not success & not crashed; counters[0..2] += number of (successes, crashes, timeouts) of this step.          */
  uint8_t *successes;         /* [N] */
  uint8_t *timeouts;          /* [N] */
  int32_t *counters;          /* [3] */
  float success_radius;
  int32_t reserved;
} AgxTaskArgs;

/* agx_env_step = agx_dynamics_substeps + the task's reward / crash / truncation / reset-set
 * in ONE launch (task may be NULL or kind NONE).                                           */
int agx_env_step(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs,
                 const float *actions_in, int k_substeps, const AgxTaskArgs *task, void *stream);

/* Which kernel instance agx_env_step launches for these arguments, as "<kernel name incl. template arguments>_<grid size
 * in threads>" (e.g. "k_env_step<4,2,false,true>_8192", "k_env_step_quad_position_32768"): the key under which profilers
 * list it (bench.py pairs its live launch time with the committed rocprofv3 counters of that instance).             */
int agx_env_step_kernel(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs, int k_substeps,
                        const AgxTaskArgs *task, char *out, int out_capacity);

/* EnvManager.compute_observations alone (env_manager.py:358-362; SURVEY 8b's minimum export set): crashes[i] |= the robot's
 * collision sphere (params->collision_radius) at its CURRENT position overlaps one of the env's obstacle boxes (buf->boxes,
 * the layout agx_boxes_from_assets writes).  The fused step accumulates the same predicate over its sub-step positions; this
 * entry point is for callers that drive simulate() / compute_observations() themselves.  No boxes bound: no-op.            */
int agx_collide_spheres_boxes(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs, void *stream);

/* BaseMultirotor.update_states alone (base_multirotor.py:287-294). */
int agx_update_states(const AgxEnvBuffers *buf, int num_envs, void *stream);

/* The elementary functions the kernels evaluate in place of torch.sin / cos / atan2 / asin / exp (utils/math.py:124-172,
 * base_lee_controller.py:136-215, the reward functions' torch.exp), on device vectors: out[i] = f(x[i]) (atan2:
 * atan2(x[i], y[i]), x = the "y" argument).  float64 inside, rounded once: correctly rounded in practice.  Diagnostic
 * entry: the parity tests compare it bit for bit with the CPU restatement.                                          */
enum { AGX_MATH_SIN = 0, AGX_MATH_COS = 1, AGX_MATH_ATAN2 = 2, AGX_MATH_ASIN = 3, AGX_MATH_EXP = 4 };
int agx_math_eval(int which, int n, const float *x, const float *y, float *out, void *stream);

/* Diagnostic: float4 streaming copy of `bytes` (multiple of 16) from src to dst -- the HBM rate a kernel of this library
 * reaches on the device at hand (bench.py's "achievable" HBM line next to the 8 TB/s of specification).            */
int agx_copy_f4(const void *src, void *dst, size_t bytes, void *stream);

/* Controller plug-in entry: BaseLeeController subclasses' update(), returning the wrench
 * [6][N] into buf->wrench_cmd (control/controllers/ *.py).  Uses buf->state/derived as is.
 * action: [N][A] row-major (clipped to +-10 like BaseMultirotor.clip_actions).          */
int agx_controller_wrench(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs,
                          const float *action, void *stream);

/* ---- robot plug-in (robots/base_robot.py:10-63, robot_manager.py:486-489) ------------------------------
 * A robot CLASS registered through robot_registry.register that overrides step(action) is called by the host once per
 * physics sub-step, like the reference's RobotManagerIGE.pre_physics_step does; what it leaves in robot_force_tensor /
 * robot_torque_tensor ([N][num_bodies][3], each body's wrench in that BODY's own frame: IGE_env_manager.py:444-449 applies
 * them with LOCAL_SPACE) is reduced to the net wrench on the rigid composite (agx_net_body_wrench) and integrated by
 * agx_env_step with AGX_CTRL_WRENCH + AGX_LAUNCH_BODY_WRENCH.
 *
 * agx_robot_step = the reference's BaseMultirotor.step(action) (base_multirotor.py:296-307) as ONE launch, for such a class
 * to call through super().step(action): update_states (derived tensors stored), clip_actions, the built-in controller,
 * ControlAllocator.allocate_output + MotorModel (motor thrusts stored; control_allocation.py:52-114), the per-body tensors
 * written like call_controller does (:246-258: every body zero, then motor link j <- force (0, 0, u_j), torque
 * (0, 0, -cq dir_j u_j); root-link mode: body_of_motor[0] <- the allocator's wrench), simulate_drag and apply_disturbance
 * accumulated into body 0 (:213-234, :260-285; buf->disturb rows of sub-step `substep`, or the device stream).          */
typedef struct AgxRobotStepArgs {
  float *force;                          /* [N][num_bodies][3] robot_force_tensor  (row-major, the reference's layout) */
  float *torque;                         /* [N][num_bodies][3] robot_torque_tensor                                      */
  int32_t num_bodies;                    /* 1 .. AGX_MAX_BODIES                                                         */
  int32_t substep;                       /* physics sub-step inside the env step (disturbance draws)                    */
  int32_t body_of_motor[AGX_MAX_MOTORS]; /* control_allocator_config.application_mask                                   */
} AgxRobotStepArgs;
int agx_robot_step(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs, const float *action /*[N][A]*/,
                   const AgxRobotStepArgs *args, void *stream);

/* Net body-frame wrench about the centre of mass of per-body wrenches given in each body's own frame:
 *   F = sum_b R_b f_b,   T = sum_b (r_b x (R_b f_b) + R_b t_b)     (b ascending, fp32, one IEEE operation per + - *)
 * rot / pos: pose of body b in the root-link frame (URDF joint origins), about the composite's centre of mass.
 * wrench_out: [N][6] row-major -- the `actions_in` of agx_env_step under AGX_LAUNCH_BODY_WRENCH.                        */
typedef struct AgxLinkFrames {
  int32_t num_bodies;
  int32_t reserved;
  float rot[AGX_MAX_BODIES][9];          /* row-major 3 x 3 */
  float pos[AGX_MAX_BODIES][3];
} AgxLinkFrames;
int agx_net_body_wrench(int num_envs, const AgxLinkFrames *frames, const float *force, const float *torque,
                        float *wrench_out, void *stream);

/* ---- tasks ----------------------------------------------------------------------
 * Position-setpoint task: compute_rewards_and_crashes + truncation test
 * (position_setpoint_task.py:205-229,245-282,172-174).  Writes reward[N], ORs the
 * distance crash into crashes, truncations = sim_steps > episode_len, and sets
 * *buf->reset_flag to 1 if any env is to be reset
 * (env_manager.py:364-371: crashes*reset_on_collision + truncations).            */
int agx_reward_position(const AgxEnvBuffers *buf, int num_envs, const float *target /*[3][N]*/,
                        int episode_len, int reset_on_collision, float *reward, void *stream);

/* process_obs_for_task (position_setpoint_task.py:194-203): obs [N][13] row-major. */
int agx_obs_position(const AgxEnvBuffers *buf, int num_envs, const float *target, float *obs,
                     void *stream);

/* Navigation task reward (navigation_task.py:416-521).  rp: 18 floats in the order of
 * config/task_config/navigation_task_config.py:30-48.  pos_err / prev_pos_err [3][N]. */
int agx_reward_navigation(const AgxEnvBuffers *buf, int num_envs, const float *target,
                          const float *rp, float curriculum_progress, float *pos_err,
                          float *prev_pos_err, int episode_len, int reset_on_collision,
                          float *reward, void *stream);

/* process_obs_for_task of the navigation task (navigation_task.py:369-393): obs [N][obs_dim]
 * row-major = unit vector to target (+0.2 U01 noise) | distance | roll, pitch (+-0.05 noise) | 0 |
 * body lin/ang velocity | robot_actions(4) | latents.  u_vec, u_euler: [N][3] U01 draws, or both
 * NULL = device generator (stream of (env, buf->step_counter)).
 * The reference fills the latents with a VAE encoding of the depth image (a conv net outside
 * the simulation hot path); here latents = grid_h x grid_w min-pooled depth image
 * (pixels [N][S][H][W], sensor 0), or left untouched when pixels == NULL.
 * min_pixel [N] (optional, num_sensors == 1 only): what agx_image_min would write
 * (post_image_reward_addition, navigation_task.py:351-357), produced by the same sweep over the
 * image.                                                                                    */
int agx_obs_navigation(const AgxEnvBuffers *buf, int num_envs, const float *target,
                       const float *u_vec, const float *u_euler, const float *pixels,
                       int num_sensors, int height, int width, int grid_h, int grid_w,
                       int obs_dim, float *obs, float *min_pixel, void *stream);

/* ---- LiDAR navigation task (task/lidar_navigation_task/lidar_navigation_task.py) -----------
 * process_image_observation (:313-363) + add_noise_to_downsampled_lidar_data (:281-310):
 * pointcloud [N][H][W][3] in the world frame (sensor 0) -> range = |p - robot_position| clipped to
 * 10 outside [0.2, 10]; time_to_collision [N] = clamp(min over rays of range / (v . dir), 0, 10);
 * pool_h x pool_w min-pooling; noise; downsampled [N][(H/pool_h) * (W/pool_w)] = 1 / range.
 * Noise: device_noise = 0 and the five tensors ([N][cells]; 0/1 masks as floats, low_* only read
 * for pooled rows >= low_row0; all may be NULL = no noise) reproduce the reference's torch draws;
 * device_noise = 1 draws from the device generator (stream of env, buf->step_counter).          */
int agx_lidar_image_obs(const AgxEnvBuffers *buf, int num_envs, int height, int width, int pool_h,
                        int pool_w, int low_row0, const float *pointcloud, const float *noise_mask,
                        const float *noise_val, const float *max_mask, const float *low_mask,
                        const float *low_val, int device_noise, float *time_to_collision,
                        float *downsampled, void *stream);

/* compute_rewards_and_crashes + compute_reward (:472-719) + truncation / reset set (:399-403, like
 * agx_reward_navigation).  action / prev_action: the task's transformed actions [N][4] row-major;
 * rp: HOST pointer to the 22 reward parameters in the order of lidar_navigation_task_config.py:30-53;
 * target [3][N], target_yaw [N], pos_err / prev_pos_err [3][N].                                  */
int agx_reward_lidar_navigation(const AgxEnvBuffers *buf, int num_envs, const float *target,
                                const float *target_yaw, const float *action,
                                const float *prev_action, const float *time_to_collision,
                                const float *rp, float curriculum_progress, float *pos_err,
                                float *prev_pos_err, int episode_len, int reset_on_collision,
                                float *reward, void *stream);

/* process_obs_for_task (:440-470): obs [N][17 + cells] = unit vector to target (+-0.1 noise) |
 * distance | roll, pitch (+-0.05 noise) | yaw error | body lin/ang velocity | robot_actions(4) |
 * downsampled.  u_vec / u_euler [N][3] U01 draws, or both NULL = device generator.              */
int agx_obs_lidar_navigation(const AgxEnvBuffers *buf, int num_envs, const float *target,
                             const float *target_yaw, const float *u_vec, const float *u_euler,
                             const float *downsampled, int cells, float *obs, void *stream);

/* *buf->step_counter_dev = (*buf->step_counter_dev + 1) mod 2^31, stream-ordered: the last node of an env step that is
 * captured into a hipGraph and replayed (small batches are launch bound: ~15 launches per navigation step).           */
int agx_step_counter_advance(const AgxEnvBuffers *buf, void *stream);

/* task_config.action_transformation_function of the reference's task configs as one launch: policy action [N][4] (clamped
 * to +-1) -> controller command.  NAV_VELOCITY: navigation_task_config.py:87-117 (speed, inclination, yaw rate -> [N][4]);
 * LIDAR_ACCELERATION: lidar_navigation_task_config.py:98-108 ([N][4]); FULLY_ACTUATED_POSE: the 7-D position + attitude
 * set-point of BASELINE configs[3] ([N][7]).  A user-supplied function (any other callable in the config) stays torch.   */
enum { AGX_ACTION_NAV_VELOCITY = 1, AGX_ACTION_LIDAR_ACCELERATION = 2, AGX_ACTION_FULLY_ACTUATED_POSE = 3 };
int agx_action_transform(int kind, int num_envs, const float *actions_in, float *out, void *stream);

/* The reset set of EnvManager.reset_terminated_and_truncated_envs (env_manager.py:364-371) from the flags as they are:
 * reset_mask = crashes * reset_on_collision | truncations, reset_flag[flag_parity] |= any.  For callers that did not
 * run one of the task reward kernels above this step (stand-alone EnvManager; tasks that set truncations in torch). */
int agx_reset_set(const AgxEnvBuffers *buf, int num_envs, int reset_on_collision, void *stream);

/* ---- task glue of the navigation-type tasks (sync-free mode: no torch launches per step) ----
 * successes / timeouts (navigation_task.py:311-326, lidar_navigation_task.py:405-418): u8 [N] each;
 * counters int32[3] += (sum successes, sum crashes, sum timeouts) for the curriculum.            */
int agx_nav_bookkeeping(const AgxEnvBuffers *buf, int num_envs, const float *target, float radius,
                        uint8_t *successes, uint8_t *timeouts, int32_t *counters, void *stream);
/* reset_idx of the tasks (navigation_task.py:166-175, lidar_navigation_task.py:164-181) for the envs
 * of buf->reset_mask: target [3][N] = bounds_min + (bounds_max - bounds_min) * U(min_ratio, max_ratio),
 * target_yaw [N] (or NULL) = U(-pi, pi), robot_prev_actions = 0 if asked.  min/max_ratio: HOST float[3];
 * u [N][4] uniform draws or NULL = device generator (env, episode).                              */
int agx_nav_target_reset(const AgxEnvBuffers *buf, int num_envs, int num_actions, const float *min_ratio,
                         const float *max_ratio, const float *u, float *target, float *target_yaw,
                         int zero_prev_actions, void *stream);
/* WarpSensor.reset_idx (warp_sensor.py:153-172) for the envs of buf->reset_mask: local_pos [N][S][3] =
 * U(min_translation, max_translation), local_quat [N][S][4] = quat_from_euler(U(min_rot, max_rot))
 * (HOST float[3] each, radians); u_pos / u_rot [N][S][3] or both NULL = device generator.         */
int agx_sensor_mount_reset(const AgxEnvBuffers *buf, int num_envs, int num_sensors,
                           const float *min_translation, const float *max_translation,
                           const float *min_rot, const float *max_rot, const float *u_pos,
                           const float *u_rot, float *local_pos, float *local_quat, void *stream);

/* ---- IMU (aerial_gym/sensors/imu_sensor.py:74-153) ------------------------------------------
 * The reference reads Isaac Gym's force sensor on the base link (total force incl. gravity, body
 * frame) every physics sub-step; here the same quantity is m * (body_force / m + R^T g).
 * agx_imu_update is called once per env step after agx_env_step with the same k: the bias random
 * walk takes k steps (z_bias [k][N][6] normal draws), noise (z_noise [N][6]) and the measurement are
 * those of the last sub-step.  z_noise == z_bias == NULL: device generator (Box-Muller on the
 * stream of (env, buf->step_counter)).  sensor_quat [N][4], bias [N][6] (in/out), imu_meas [N][6]
 * = [accel(3), gyro(3)] clamped to +-max_value.                                                 */
typedef struct AgxImuArgs {
  float bias_std[6], noise_std[6], max_value[6], max_bias_init[6];
  float min_rot[3], max_rot[3]; /* sensor mount perturbation, radians                              */
  float g_world[3];             /* gravity * (1 - gravity_compensation)                            */
  float sqrt_dt, mass;
  int32_t world_frame, enable_noise, enable_bias;
} AgxImuArgs;
int agx_imu_update(const AgxEnvBuffers *buf, int num_envs, int k_substeps, const AgxImuArgs *args,
                   const float *sensor_quat, const float *z_noise, const float *z_bias, float *bias,
                   float *imu_meas, void *stream);
/* IMUSensor.reset_idx for the envs of buf->reset_mask (when reset_flag[flag_parity] != 0):
 * bias = max_bias_init * (2 (u - 0.5)), sensor_quat = quat_from_euler(U(min_rot, max_rot)).
 * u_bias [N][6], u_rot [N][3] uniform draws, or both NULL = device generator (env, episode).     */
int agx_imu_reset(const AgxEnvBuffers *buf, int num_envs, const AgxImuArgs *args, const float *u_bias,
                  const float *u_rot, float *bias, float *sensor_quat, void *stream);

/* ---- reset ----------------------------------------------------------------------
 * Masked reset, in the order of EnvManager.reset_idx (env_manager.py:273-301):
 *   env bounds (IsaacGymEnv.reset_idx, IGE_env_manager.py:513-519), robot state
 *   (BaseMultirotor.reset_idx base_multirotor.py:177-205), controller gains
 *   (BaseLeeController.randomize_params base_lee_controller.py:101-118, if enabled),
 *   motor model (MotorModel.reset_idx motor_model.py:140-154), sim_steps[reset] = 0;
 * and, when *buf->reset_flag != 0, update_states for ALL envs (base_multirotor.py:205):
 * the reference refreshes every env's derived tensors whenever at least one env resets.
 * Nothing is touched when the flag is 0.
 * The uniform draws are inputs, in the AoS layout torch produces them:
 *   u_bounds_lo/hi [N][3], u_state [N][13], u_gains [N][12],
 *   u_tau_inc/u_tau_dec/u_thrust/u_kT [N][M].
 * If u_state is NULL the kernel draws them itself with a counter-based generator
 * (Philox4x32-10 keyed by `seed`, counter = (env, episode_count[env], stream, block)):
 * the sync-free mode, reproducible (the parity tests restate the generator bit for bit).
 * randomize_gains != 0 resamples the controller gains (randomize_params).
 * The reset set is buf->reset_mask.                                                   */
typedef struct AgxResetArgs {
  const float *u_bounds_lo, *u_bounds_hi;
  const float *u_state, *u_gains, *u_tau_inc, *u_tau_dec, *u_thrust, *u_kT;
  float lower_bound_min[3], lower_bound_max[3], upper_bound_min[3], upper_bound_max[3];
  float min_state[13], max_state[13];
  float gains_min[12], gains_max[12];
  float tau_inc_min, tau_inc_max, tau_dec_min, tau_dec_max, kT_min, kT_max;
  int32_t randomize_gains;
  uint64_t seed;
} AgxResetArgs;

int agx_reset_masked(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs,
                     const AgxResetArgs *args, void *stream);

/* The robot side of a navigation task's step as ONE launch (sync-free mode: device generator only): agx_reset_masked, then for
 * the envs that reset agx_sensor_mount_reset and agx_nav_target_reset, then agx_sensor_pose for every env -- the same device
 * functions as those four entry points, in that order (the mount and the target draws are keyed by the episode count the robot
 * reset has just advanced).  Replaces four dependent 5-us launches at the 256 .. 1024 envs an RL run uses.
 * num_sensors = 0: no sensor part; randomize_mount = 0: the mount is left alone; reset_target = 0: no target part.   */
typedef struct AgxNavRobotSideArgs {
  int32_t num_sensors, randomize_mount;
  float mount_t_min[3], mount_t_max[3], mount_r_min[3], mount_r_max[3];
  float *local_pos, *local_quat;   /* [N][S][3], [N][S][4]: the mount in the robot frame */
  float frame_quat[4];
  float *sensor_pos, *sensor_quat; /* out [N][S][3], [N][S][4]: world pose of every sensor */
  int32_t reset_target, num_actions, zero_prev_actions, pad_;
  float target_ratio_min[3], target_ratio_max[3];
  float *target;      /* [3][N] */
  float *target_yaw;  /* [N] or NULL */
} AgxNavRobotSideArgs;
int agx_nav_robot_side(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs, const AgxResetArgs *args,
                       const AgxNavRobotSideArgs *nav, void *stream);

/* agx_reset_masked + agx_obs_position in one launch (the position task has no sensor to
 * render between reset and observation).                                                  */
int agx_post_step_position(const AgxRobotParams *params, const AgxEnvBuffers *buf, int num_envs,
                           const AgxResetArgs *args, const float *target, float *obs, void *stream);

/* One task.step() of the position-setpoint task as a single host call (two launches:
 * agx_env_step with the position epilogue, then agx_post_step_position).  Toggles
 * buf->flag_parity first, exactly like the host does once per env step.  `plan` only bundles
 * arguments the caller would otherwise pass to those two entry points; nothing is retained.   */
typedef struct AgxPositionStepPlan {
  const AgxRobotParams *params;
  AgxEnvBuffers *buf;
  const AgxTaskArgs *task;
  const AgxResetArgs *reset;
  const float *target; /* [3][N] */
  float *obs;          /* [N][13] */
  int32_t num_envs;
  int32_t k_substeps;
} AgxPositionStepPlan;
int agx_position_task_step(const AgxPositionStepPlan *plan, const float *actions_in, void *stream);

/* ---- the same step in the reference-faithful RNG mode (args={"strict_rng": True}) -------------------------------------
 * The reference consumes torch's generator only on steps on which some env resets (`if len(env_ids) > 0` behind a
 * nonzero(): env_manager.py:364-375), with one rand_like per reset quantity (IGE_env_manager.py:513-519,
 * base_multirotor.py:177-205, motor_model.py:140-154).  Keeping that contract costs the host one bit per step.
 *
 * agx_torch_uniform_fill: the numbers `Tensor.uniform_(0, 1)` produces for `count` dense float32 tensors called one after
 * another on a device generator whose state is (seed, offset) -- torch 2.10 / ROCm 7: ATen/native/cuda/DistributionTemplates.h
 * (launch policy, thread -> element mapping, bound reversal) over rocrand's Philox4x32-10 (csrc/agx_strict.hip cites the
 * lines) -- in ONE launch.  sm_count / max_threads_per_sm: the device properties torch's launch policy reads
 * (multiProcessorCount, maxThreadsPerMultiProcessor).  *offset_after = the generator offset after those calls
 * (Generator.set_offset).  Pinned bit for bit against the real calls by tests/test_gpu_strict_fast.py.
 *
 * agx_host_word_create: one 32-bit word of mapped, coherent host memory (hipHostMalloc) that kernels store to and the host
 * polls; agx_position_task_step_strict: agx_position_task_step with, between its two launches, a one-lane kernel that
 * publishes (sequence << 1 | reset_flag[parity]) into that word, a host spin until it arrives (no stream synchronisation),
 * and -- only when the flag is set -- the uniform fill of the reset's draw tensors (plan->reset->u_*).  *drew = the flag;
 * *offset_after = the generator offset the caller must set (unchanged when nothing was drawn).                              */
#define AGX_MAX_UNIFORM_SEGMENTS 8
int agx_torch_uniform_fill(int count, float *const *out, const int64_t *numel, uint64_t seed, uint64_t offset, int sm_count,
                           int max_threads_per_sm, uint64_t *offset_after, void *stream);
int agx_host_word_create(uint32_t **word);
int agx_host_word_destroy(uint32_t *word);
typedef struct AgxStrictStepPlan {
  const AgxPositionStepPlan *plan;
  uint32_t *host_word;                       /* agx_host_word_create */
  int32_t count;                             /* draw tensors, in the reference's call order */
  int32_t timeout_ms;                        /* 0: 10 s */
  float *out[AGX_MAX_UNIFORM_SEGMENTS];
  int64_t numel[AGX_MAX_UNIFORM_SEGMENTS];
  uint64_t seed, offset;                     /* the generator's state before this step (Generator.initial_seed / get_offset) */
  int32_t sm_count, max_threads_per_sm;
} AgxStrictStepPlan;
int agx_position_task_step_strict(const AgxStrictStepPlan *plan, const float *actions_in, int *drew, uint64_t *offset_after,
                                  void *stream);

/* Obstacle pose randomisation of the reset envs: AssetManager.reset_idx
 * (asset_manager.py:51-71) incl. the half-obstacle resample of env_manager.py:283-295.
 * u1/u2 [N][K][13] and u_sel [N] are the uniform draws (first / second rand_like and the
 * bernoulli(0.15)); all NULL = device generator.  u_bounds as in AgxResetArgs (the new env
 * bounds are needed before the robot reset kernel has stored them).  Assets with index
 * >= the active count are parked at -1000 m.                                              */
int agx_reset_assets(const AgxEnvBuffers *buf, int num_envs, int num_assets,
                     const AgxResetArgs *args, const float *u1, const float *u2,
                     const float *u_sel, const float *min_ratio, const float *max_ratio,
                     int num_obstacles, int num_keep_in_env, float *asset_state, void *stream);

/* ---- scene / ray-cast -------------------------------------------------------------
 * Scene = per-env triangle soup with a fixed topology: T triangles, each owned by one
 * asset (obstacle) whose pose comes from env_asset_state_tensor.                      */

/* WarpEnv.reset_idx (warp_env_manager.py:40-54): v_world = tf_apply(q_asset, p_asset, v).
 * tri_local/tri_world: [N][T][9] (a,b,c xyz); tri_asset: [T]; asset_state: [N][K][13].
 * Only envs with mask[env] != 0 are transformed (mask NULL = all).                    */
int agx_scene_transform(int num_envs, int num_tris, int num_assets, const float *tri_local,
                        const int32_t *tri_asset, const float *asset_state, const uint8_t *mask,
                        float *tri_world, void *stream);

/* Multi-primitive assets (URDFs with several links, e.g. the reference's `trees`): prim_state [N][P][13]
 * pose = asset pose (x) local pose (local_pos [N][P][3], local_quat [N][P][4], prim_asset [N][P] = owning
 * asset index in that env), velocities copied.  prim_state then stands in for asset_state (num_assets = P) in
 * agx_scene_transform (tri_asset = primitive index) and agx_boxes_from_assets.                     */
int agx_prims_from_assets(int num_envs, int num_prims, int num_assets, const int32_t *prim_asset,
                          const float *asset_state, const float *local_pos, const float *local_quat,
                          const uint8_t *mask, float *prim_state, void *stream);

/* Kinematic obstacles: EnvManager.step(actions, env_actions) with env_actions = obstacle twists
 * (obstacle_manager.py:40-44, examples/dynamic_env_example.py:33-45).  twist [N][K][6] = world-frame
 * linear and angular velocity; asset_state [N][K][13] gets the twist in its velocity slots and its
 * pose advanced by k sub-steps of dt (same rule as the robot integrator).  Follow with
 * agx_scene_transform / agx_bvh_build / agx_boxes_from_assets (mask NULL) to move the geometry.  */
int agx_assets_integrate(int num_envs, int num_assets, float *asset_state, const float *twist,
                         float dt, int k_substeps, void *stream);

/* wp.Mesh(...) BVH build / mesh.refit() (warp_env_manager.py:162-166, 52-53).
 * A workgroup builds a binary LBVH over the T triangles of an env in LDS and stores
 * T-1 nodes of 16 floats each: [lo_l(3) child_l | hi_l(3) child_r | lo_r(3) pad | hi_r(3) pad],
 * child < 0 encodes a leaf: triangle index = ~child.
 * prims_per_object > 0: the soup is made of objects of that many consecutive triangles (boxes:
 * 12); the Morton order is then taken over object centres, which keeps an object's triangles
 * together (two-level hierarchy).  0 = order by triangle centroid.
 * mask != NULL rebuilds only the flagged envs and needs `work` (int32[num_envs + 2], scratch):
 * the dirty env ids are compacted into it and a CU-sized persistent grid pulls from the list.
 * With objects, the order of the keys is obtained from a sort of the OBJECTS' codes plus every triangle's rank inside its
 * object (the full key sort runs only when two objects that are in the env share a code); AGX_BVH_FULL_SORT ORed into
 * prims_per_object forces the full sort -- a test hook: both must give the same tree.   */
#define AGX_BVH_FULL_SORT 0x40000000
/* AGX_BVH_BOX_OBJECTS ORed into prims_per_object (= 12): every object is a box of trimesh.creation.box's topology (the
 * reference's box URDFs through urdfpy: 8 corners, the 12 faces in trimesh's order).  The builder then ends the tree at the
 * OBJECT: the subtree root whose leaves are exactly one object's 12 triangles becomes an OBJECT NODE -- its 64-byte record
 * holds the box's frame instead of two child boxes,
 *     [0..2] axis x (unit, world)  [3] half extent x     [4..6] axis y  [7] half extent y
 *     [8..10] axis z               [11] half extent z    [12..14] centre  [15] index of the object's first triangle (int bits)
 * (derived from the object's world-frame triangles: corners 0, 1, 2, 4 of the box), and the parent's child reference carries
 * AGX_BVH_OBJECT_REF.  The ray-cast kernels intersect the box's slabs in its own frame, conservatively (tolerance >> rounding),
 * and run the exact triangle test only on the face(s) a ray can enter through: the closest hit over the triangles the exact test
 * accepts -- the result's definition -- is unchanged, the five in-object nodes and most face tests are gone.  An object whose
 * triangles do not make an orthogonal box (or whose keys interleave with another object's) keeps its triangle subtree.
 * AGX_BVH_OBJECT_TREE (round 6; with AGX_BVH_BOX_OBJECTS, 2 <= T / 12 <= 256, no AGX_BVH_FULL_SORT): the tree is BUILT over the
 * K = T / 12 objects instead of over the triangles: records 0 .. K - 2 are the radix tree over the objects' Morton keys (root 0),
 * records K - 1 + 5 o .. + 4 belong to object o -- its object node at the first of them, or, for an object that is not a recognised
 * box (parked beyond the curriculum level at -1000 m, or simply not a box), a five-node subtree over its six triangle pairs with its
 * root at the fifth.  Consumers follow child references only; nothing depends on the numbering.  For scenes whose objects ARE boxes
 * (one dirty env: 24 us instead of 55-60); the quick subtrees of non-box chunks (cylinders, spheres, meshes) traverse 7-9 % slower
 * than their LBVH subtrees (profiles/forest_probe_r06.py), so callers leave the flag off for such scenes.                       */
#define AGX_BVH_BOX_OBJECTS 0x20000000
#define AGX_BVH_OBJECT_TREE 0x10000000
#define AGX_BVH_OBJECT_REF 0x40000000   /* bit 30 of a non-negative child reference: the child is an object node */
size_t agx_bvh_nodes_bytes(int num_envs, int num_tris);
int agx_bvh_build(int num_envs, int num_tris, int prims_per_object, const float *tri_world,
                  const uint8_t *mask, float *nodes, int32_t *work, void *stream);

/* Obstacle OBBs for the collision test, from the same asset poses:
 * boxes [K][11][N] <- asset_state [N][K][13], half_extents [N][K][3].                  */
int agx_boxes_from_assets(int num_envs, int num_assets, const float *asset_state,
                          const float *half_extents, const uint8_t *mask, float *boxes,
                          void *stream);

/* The three calls above as ONE, for the geometry refresh behind a reset (asset_manager.py:51-71 moves the obstacles of the
 * envs that reset; env_manager.py:283-295; the reference then refits its Warp meshes: warp_env.py / `mesh.refit()`):
 * world-frame triangles, collision boxes (boxes may be NULL) and the tree of the envs flagged in `mask`.  With a mask: the
 * dirty env ids are compacted into `work` and ONE persistent launch transforms, boxes and builds per dirty env -- as three
 * launches the two small ones are a dispatch over every env each (22 + 9 us per step at 8192 envs x 106 obstacles for a few
 * dozen dirty envs).  mask == NULL: every env, through the three stand-alone kernels.  Same arithmetic either way.          */
int agx_scene_refresh(int num_envs, int num_tris, int num_assets, const float *tri_local,
                      const int32_t *tri_asset, const float *asset_state, const float *half_extents,
                      int prims_per_object, const uint8_t *mask, float *tri_world, float *boxes,
                      float *nodes, int32_t *work, void *stream);

/* AssetManager.reset_idx AND the geometry refresh behind it for the envs of buf->reset_mask (agx_reset_assets + agx_scene_refresh,
 * asset_manager.py:51-71 -> warp_env_manager.py:40-54), draws from the device generator (args->u_* NULL).  Up to 2048 envs this is
 * ONE launch -- a workgroup per env that resets the env's obstacle poses, moves its triangles and collision boxes and rebuilds its
 * tree; a clean env's workgroup leaves at once, a step without a reset costs one dispatch -- above, the three launches it replaces
 * at small batches (asset reset, mask compaction, persistent refresh).  num_assets = obstacles per env (one rigid piece each:
 * multi-primitive scenes keep agx_reset_assets + agx_prims_from_assets + agx_scene_refresh).  Same device functions either way. */
int agx_scene_reset_refresh(const AgxEnvBuffers *buf, int num_envs, int num_tris, int num_assets, const AgxResetArgs *args,
                            const float *min_ratio, const float *max_ratio, int num_obstacles, int num_keep, float *asset_state,
                            const float *tri_local, const int32_t *tri_asset, const float *half_extents, int prims_per_object,
                            float *tri_world, float *boxes, float *nodes, int32_t *work, void *stream);

/* WarpSensor.update pose composition (warp_sensor.py:177-187).
 * local_pos [N][S][3], local_quat [N][S][4], frame_quat [4] -> pos [N][S][3], quat [N][S][4] */
int agx_sensor_pose(const AgxEnvBuffers *buf, int num_envs, int num_sensors,
                    const float *local_pos, const float *local_quat, const float *frame_quat,
                    float *pos, float *quat, void *stream);

enum {
  AGX_RAY_RANGE = 0, AGX_RAY_DEPTH = 1, AGX_RAY_POINTCLOUD = 2, AGX_RAY_POINTCLOUD_WORLD = 3,
  /* draw_optimized_kernel_normal_faceID (warp_camera_kernels.py:70-121, warp_lidar_kernels.py:90-126):
   * pixels [..][3] = geometric normal of the hit face in the sensor / world frame (0 on a miss),
   * seg = face index (-1 on a miss; tri_seg is not read)                                          */
  AGX_RAY_NORMAL = 4, AGX_RAY_NORMAL_WORLD = 5
};

/* WarpSensor.apply_range_limits + normalize_observation (warp_sensor.py:216-247) as the epilogue of the
 * ray-cast itself.  With `limits` != NULL (scalar images only: modes RANGE / DEPTH) a pixel is stored
 * already limited and normalised -- bit for bit what agx_sensor_postprocess(count, pixels, NULL, NULL,
 * ...) leaves -- and the image needs no second pass over HBM.  NULL = raw distances (and the only
 * choice with sensor noise, which the reference applies BEFORE the limits).                          */
typedef struct AgxRangeLimits {
  float min_range, max_range;
  float far_oor, near_oor; /* cfg.far_out_of_range_value, cfg.near_out_of_range_value */
  int32_t normalize;       /* cfg.normalize_range: p / max_range                     */
} AgxRangeLimits;

/* DepthCameraWarpKernels.draw_optimized_kernel_{depth_range,depth_range_segmentation,
 * pointcloud,pointcloud_segmentation} (warp_camera_kernels.py:176-282, 13-66, 125-172).
 * kinv = {K_inv[0][0], K_inv[0][2], K_inv[1][1], K_inv[1][2]} (warp_cam.py:31-64).
 * pixels [N][S][H][W] (or x3), seg [N][S][H][W] int32 or NULL; tri_seg [N][T] int32.    */
int agx_raycast_camera(int num_envs, int num_sensors, int width, int height, const float *kinv,
                       float far_plane, int c_x, int c_y, int mode, const float *cam_pos,
                       const float *cam_quat, const float *tri_world, const int32_t *tri_seg,
                       const float *nodes, int num_tris, float *pixels, int32_t *seg,
                       const AgxRangeLimits *limits, void *stream);

/* StereoCameraWarpKernels.* (warp_stereo_camera_kernels.py:13-299): as agx_raycast_camera
 * (modes 0..3), but a pixel is valid only if the stereo partner at cam_pos + R(cam_quat)
 * (-baseline, 0, 0) also sees the hit point (second, any-hit ray from 0.999 t).  Occluded:
 * -1 / seg -2; a missed ray whose far-plane point the partner sees: 1000, else -1.            */
int agx_raycast_stereo_camera(int num_envs, int num_sensors, int width, int height,
                              const float *kinv, float far_plane, float baseline, int c_x, int c_y,
                              int mode, const float *cam_pos, const float *cam_quat,
                              const float *tri_world, const int32_t *tri_seg, const float *nodes,
                              int num_tris, float *pixels, int32_t *seg,
                              const AgxRangeLimits *limits, void *stream);

/* LidarWarpKernels.draw_optimized_kernel_{range,range_segmentation,pointcloud,
 * pointcloud_segmentation} (warp_lidar_kernels.py:167-194,130-163,13-86).
 * ray_vectors [H][W][3] (warp_lidar.py:40-64).                                          */
int agx_raycast_lidar(int num_envs, int num_sensors, int width, int height,
                      const float *ray_vectors, float far_plane, int mode, const float *pos,
                      const float *quat, const float *tri_world, const int32_t *tri_seg,
                      const float *nodes, int num_tris, float *pixels, int32_t *seg,
                      const AgxRangeLimits *limits, void *stream);

/* Which kernel instance the three entry points above launch for these sizes, as "<kernel name incl. template arguments>_<grid
 * size in threads>" (e.g. "k_raycast<false,0>_6291456"): the key under which profilers list it (like agx_env_step_kernel).
 * variant: 0 depth / range / point cloud, 1 normal + face id, 2 stereo.                                                  */
int agx_raycast_kernel(int num_envs, int num_sensors, int width, int height, int lidar, int variant, char *out, int out_len);

/* WarpSensor.apply_noise / apply_range_limits / normalize_observation
 * (warp_sensor.py:202-247), scalar images, in place.  z_normal/u_dropout optional.     */
int agx_sensor_postprocess(size_t count, float *pixels, const float *z_normal,
                           const float *u_dropout, float std_a, float std_b, float std_c,
                           float mean_offset, float dropout_prob, float min_range,
                           float max_range, float far_oor, float near_oor, int normalize,
                           void *stream);

/* Point-cloud branch of the same three functions: pixels [count][3]; noise / dropout act on
 * every component, the range limits on the point's norm (all three components replaced).
 * limits = 0 for world-frame clouds, which the reference neither limits nor normalises.     */
int agx_sensor_postprocess_points(size_t count, float *pixels, const float *z_normal,
                                  const float *u_dropout, float std_a, float std_b, float std_c,
                                  float mean_offset, float dropout_prob, float min_range,
                                  float max_range, float far_oor, float near_oor, int limits,
                                  int normalize, void *stream);

/* NavigationTask.post_image_reward_addition's min over the image
 * (navigation_task.py:351-357): min_pixel[N] = min(10*img, with img<0 -> 10).          */
int agx_image_min(int num_envs, int pixels_per_env, const float *pixels, float *min_pixel,
                  void *stream);

/* ---- sharded stepping: the per-step observation exchange -----------------------
 * One RCCL all-gather per env step of the [N_local][obs_dim + 3] fp32 rows the observation kernels
 * write (AgxEnvBuffers.step_rows), enqueued by a worker thread of this library on its own stream so
 * that the gather of step t overlaps the kernels of step t+1 and costs the stepping thread one
 * event record + one stream wait.  The reference has no distributed path (replicas only), so this
 * replaces nothing upstream; the host mirror is sharding.StepGather.
 *   rccl_path: the librccl.so to bind at run time (the one torch loaded); NULL = "librccl.so".
 *   agx_exchange_unique_id: rank 0 only; the 128 id bytes travel to the other ranks out of band
 *                           (torch.distributed broadcast in the host mirror).
 *   agx_exchange_create:    collective over all ranks (ncclCommInitRank), `device` = HIP ordinal.
 *   agx_exchange_post:      rows `send` [count] of this step (written on `stream`) -> `recv`
 *                           [world * count]; returns at once.  One post per parity in flight.
 *                           signal = AgxEnvBuffers.step_signal of the env whose kernels write `send`
 *                           and seq = its step_counter + 1 for this step: the gather waits for
 *                           signal[parity] >= seq on the device (no HIP call on this thread);
 *                           signal = NULL: an event recorded on `stream` now orders the gather.
 *   agx_exchange_probe:     must return 1 before a post with signal != NULL: checks (and arranges) that
 *                           the communication stream does not share a hardware queue with `stream`,
 *                           where a device-side wait would block its own producer; 0 = use events.
 *   agx_exchange_wait:      `stream` waits (no host block beyond the enqueue hand-off) for the
 *                           latest posted gather of `parity`: after it the stream may read that
 *                           recv buffer and overwrite that send buffer.
 *   agx_exchange_step:      post(parity) then wait(wait_parity) in one call (wait_parity < 0: none).
 *   agx_exchange_info:      rank and size AS THE COMMUNICATOR REPORTS THEM (ncclCommUserRank / ncclCommCount):
 *                           lets the caller assert that the collective really spans the ranks it was built for. */
typedef struct AgxExchange AgxExchange;
int agx_exchange_unique_id(const char *rccl_path, void *id_out, int id_bytes);
int agx_exchange_create(const char *rccl_path, const void *id, int id_bytes, int rank, int world,
                        int device, AgxExchange **out);
int agx_exchange_post(AgxExchange *x, int parity, const float *send, float *recv,
                      size_t count_per_rank, const uint32_t *signal, uint32_t seq, void *stream);
int agx_exchange_probe(AgxExchange *x, void *stream);
int agx_exchange_wait(AgxExchange *x, int parity, void *stream);
int agx_exchange_step(AgxExchange *x, int parity, const float *send, float *recv,
                      size_t count_per_rank, const uint32_t *signal, uint32_t seq, int wait_parity,
                      void *stream);
int agx_exchange_info(AgxExchange *x, int *rank, int *world);
int agx_exchange_destroy(AgxExchange *x);

/* Peer push (round 3): the same exchange WITHOUT a collective kernel per step.  Every rank stores its rows straight into
 * every peer's receive buffer (peer memory mapped through hipIpcMemHandle: one xGMI link per destination) and raises a
 * per-sender arrival flag there; the consumer's stream waits on its own flags.  post / wait / step / probe / info / destroy
 * are the entry points above (post ignores `recv`: the receive buffer belongs to the exchange; wait launches a one-wave flag
 * wait on `stream` instead of a cross-queue event wait).
 *   agx_exchange_create_push:  allocates this rank's receive buffer [slots][world][count_per_rank] and flags; not collective.
 *   agx_exchange_push_export:  2 x 64 bytes (hipIpcMemHandle of buffer and flags) for the peers; the host mirror all-gathers
 *                              them through torch.distributed.
 *   agx_exchange_push_connect: maps the peers' buffers (world x 128 bytes, rank-major); after it posts are allowed.
 *   agx_exchange_push_buffer:  this rank's receive buffer and slot count: the rows gathered by the s-th post (s = 1, 2, ...)
 *                              lie in slot (s - 1) % slots once agx_exchange_wait for that post's parity has passed.       */
int agx_exchange_create_push(int rank, int world, int device, size_t count_per_rank, AgxExchange **out);
int agx_exchange_push_export(AgxExchange *x, void *handles_out, int bytes);
int agx_exchange_push_connect(AgxExchange *x, const void *all_handles, int bytes);
int agx_exchange_push_buffer(AgxExchange *x, void **recv, int *slots, int *flags_uncached);
/* Rows pushed by the observation kernels THEMSELVES (AgxEnvBuffers.push_*: no launch and no host call per step for the
 * exchange): agx_exchange_push_peers hands out the addresses, in this process, of every rank's receive buffer and flag array
 * ([world] each, own included) and the device-visible time-out word; agx_exchange_push_wait_seq makes `stream` wait until
 * the rows with sequence number `seq` of every rank have arrived here.                                                   */
/* Peer push by the kernels: a new env step -- next sequence number, its slot (step_rows), the step (two back) to wait for.
 * agx_position_task_step calls it itself; callers of agx_env_step call it once per env step before the launch.           */
int agx_push_advance(AgxEnvBuffers *buf);
int agx_exchange_push_peers(AgxExchange *x, void **recv_out, void **flags_out, void **timed_out);
int agx_exchange_push_wait_seq(AgxExchange *x, uint32_t seq, void *stream);
/* Has anything gone wrong so far (worker thread; a bounded device-side wait that gave up)?  Two host reads.            */
int agx_exchange_check(AgxExchange *x);
/* Connection self-test of the peer push; every rank calls it after agx_exchange_push_connect and before the first post (the
 * callers put a barrier of their own behind it).  Each rank stores one word into every rank's receive buffer through the
 * mapped addresses -- the way the observation kernels store rows --, raises a flag, waits at most timeout_ms for every rank's
 * flag and compares what arrived.  *passed = 1: this rank received every rank's word.  The buffers are left as found. */
int agx_exchange_push_selftest(AgxExchange *x, int timeout_ms, int *passed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AERIAL_GYM_HIP_H */
