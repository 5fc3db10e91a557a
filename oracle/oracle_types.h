/*
 * TEST INFRASTRUCTURE -- CPU oracle for the Aerial Gym per-env simulation step.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call anything in oracle/.  The product package
 * (aerial_gym_simulator_amd/) never does.
 *
 * Plain C, fp32, compiled with -ffp-contract=off so every +,-,*,/,sqrtf is a
 * single correctly-rounded IEEE operation (needed for the bit-exact gates:
 * crash flags, segmentation ids, depth).
 *
 * Layout here deliberately follows the REFERENCE (AoS, row-major [N, C]
 * tensors, quaternions xyzw) -- not the product's SoA layout -- so that the
 * oracle can be compared 1:1 with tensors produced by the reference's own
 * torch code (tests/golden/).
 */
#ifndef ORACLE_TYPES_H
#define ORACLE_TYPES_H

#include <stdint.h>

#define ORC_MAX_MOTORS 8

/* controller ids; names follow aerial_gym/control/__init__.py:42-100 */
enum {
  ORC_CTRL_NONE = 0,          /* no_control: action = motor thrust references          */
  ORC_CTRL_POSITION = 1,      /* lee_position_control     position_control.py:20-51     */
  ORC_CTRL_VELOCITY = 2,      /* lee_velocity_control     velocity_control.py:18-51     */
  ORC_CTRL_ATTITUDE = 3,      /* lee_attitude_control     attitude_control.py:16-43     */
  ORC_CTRL_RATES = 4,         /* lee_rates_control        rates_control.py:16-30        */
  ORC_CTRL_ACCELERATION = 5,  /* lee_acceleration_control acceleration_control.py:16-45 */
  ORC_CTRL_VEL_STEERING = 6,  /* velocity_steeing_angle_controller.py:15-45             */
  ORC_CTRL_FULLY_ACTUATED = 7 /* fully_actuated_control.py:14-32 (7-D action)           */
};

/* Robot / sim constants shared by all envs (the reference asserts one robot
 * model per simulation: robots/robot_manager.py:449-456).                   */
typedef struct {
  int32_t num_motors;     /* M                                                */
  int32_t num_actions;    /* A                                                */
  int32_t controller;     /* ORC_CTRL_*                                       */
  int32_t root_link_mode; /* control_allocator_config.force_application_level == "root_link" */
  float dt;               /* sim.dt, base_sim_config.py:21                    */
  float dt_over_6;        /* float(dt / 6.0), dt as the python double: motor_model.py:198 forms the RK4 weight
                             from python scalars (float(dt) / 6.0f is 1 ulp away for dt = 0.01)          */
  float gravity[3];       /* base_sim_config.py:23                            */
  float mass;             /* composite, robot_manager.py:295-435              */
  float inertia[9];       /* composite J about COM, body frame, row-major     */
  float inertia_inv[9];
  float alloc[6 * ORC_MAX_MOTORS];      /* A  (6 x M) row-major, cfg.allocation_matrix     */
  float alloc_pinv[ORC_MAX_MOTORS * 6]; /* A+ (M x 6) row-major, control_allocation.py:38 */
  float wrench_map[6 * ORC_MAX_MOTORS]; /* body wrench per unit thrust of motor i when the
                                           force is applied at the motor LINK (URDF frames) */
  float motor_dir[ORC_MAX_MOTORS];
  float cq;               /* thrust_to_torque_ratio                           */
  /* motor model, control/motor_model.py */
  int32_t use_rps;
  int32_t use_discrete_approximation;
  int32_t integration_rk4;
  float min_thrust, max_thrust, max_rate;
  /* Lee controller */
  float max_yaw_rate;     /* lee_controller_config.py:21                      */
  /* body drag, base_multirotor.py:260-285 */
  float lin_drag_linear[3], lin_drag_quadratic[3];
  float ang_drag_linear[3], ang_drag_quadratic[3];
  /* rigid-body integrator (PhysX actor options, base_quad_config.py:87-97) */
  float linear_damping, angular_damping;
  float max_linear_velocity, max_angular_velocity;
  /* collision sphere (quad.urdf:16) */
  float collision_radius;
} OrcRobotParams;

#endif
