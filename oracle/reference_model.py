"""Independent pure-Python transliteration of the reference's particle step.

TEST INFRASTRUCTURE ONLY (see oracle/eggsim_oracle.h).  This file is the *second*,
independently structured restatement used to pin oracle/eggsim_oracle.c: it keeps
the reference's own data layout (one flat array per particle type with stride 20,
1-based particle indices, Lua-style tables as dicts/lists, Szudzik pairing keys)
where the C oracle uses SoA arrays and integer-pair keys.  Both must agree bit for
bit; tests/golden/*.npz are generated from THIS model by oracle/gen_golden.py.

Parity unpinned against the running reference: no Lua interpreter exists in this
pipeline, so neither restatement can be compared with the real thing.

Citations: L = /root/reference/simulation_handler.lua, M = /root/reference/math.lua,
D = /root/reference/simulation_handler_default_config.lua.
"""
import math

EPS = 1e-8  # M:2
INF = float("inf")

# L:713-735 (offsets are added to a 1-based base index, as in the reference)
X, Y, Z, VX, VY, PX, PY, RADIUS, MASS_T, MASS, INV_MASS, CELL_X, CELL_Y, BATCH_ID = range(14)
R_, G_, B_, A_, LAST_X, LAST_Y = 14, 15, 16, 17, 18, 19
STRIDE = LAST_Y + 1


def offset(particle_i):  # L:738-740
    return (particle_i - 1) * STRIDE + 1


def clamp(x, lo, hi):  # M:16-26
    if x < lo:
        x = lo
    if x > hi:
        x = hi
    return x


def mix(lo, hi, t):  # M:33-35
    return lo * (1 - t) + hi * t


def normalize(x, y):  # M:53-60
    magnitude = math.sqrt(x * x + y * y)
    if magnitude < EPS:
        return 0.0, 0.0
    return x / magnitude, y / magnitude


def magnitude(x, y):  # M:66-68
    return math.sqrt(x * x + y * y)


def distance(x1, y1, x2, y2):  # M:96-100
    return magnitude(x2 - x1, y2 - y1)


def squared_distance(x1, y1, x2, y2):  # M:108-112
    dx = x2 - x1
    dy = y2 - y1
    return dx * dx + dy * dy


def xy_to_hash(x, y):  # L:1474-1483
    a = (x * 2) if x >= 0 else (-x * 2 - 1)
    b = (y * 2) if y >= 0 else (-y * 2 - 1)
    if a >= b:
        return a * a + a + b
    return b * b + a


def default_configs():  # D:1-70, solver-relevant keys
    white = dict(damping=0.1, follow_strength=1 - 0.004, cohesion_strength=1 - 0.2,
                 cohesion_interaction_distance_factor=2, collision_strength=1 - 0.0025,
                 collision_overlap_factor=2, min_mass=1, max_mass=1 * 1.8, min_radius=4, max_radius=4)
    yolk = dict(damping=0.1, follow_strength=1 - 0.004, cohesion_strength=1 - 0.002,
                cohesion_interaction_distance_factor=3, collision_strength=1 - 0.001,
                collision_overlap_factor=2, min_mass=1, max_mass=1 * 1.35, min_radius=4, max_radius=4)
    return white, yolk


class Array(dict):
    """A Lua table used as a 1-based array: index -> value, `#t` tracked by hand."""
    pass


class ReferenceModel:
    def __init__(self, white_config=None, yolk_config=None):  # L:425-459
        if white_config is None:
            white_config, yolk_config = default_configs()
        if yolk_config is None:
            yolk_config = white_config
        self._white_config = dict(white_config)
        self._yolk_config = dict(yolk_config)
        self._mass_distribution_variance = 4
        self._max_collision_fraction = 0.05
        # _reinitialize, L:465-563
        self._batch_id_to_batch = {}
        self._current_batch_id = 1
        self._n_batches = 0
        self._white_data = []  # python list, index 0 unused so that indices stay 1-based
        self._white_data.append(None)
        self._total_n_white_particles = 0
        self._yolk_data = [None]
        self._total_n_yolk_particles = 0
        self._elapsed = 0
        self._interpolation_alpha = 0
        self._last_white_env = None
        self._last_yolk_env = None
        self.pass_log = []  # (which, sub_step, pass, n_collided, cut)
        self.n_steps = 0
        self._step(0, 1, 1)  # L:562
        self.n_steps = 0

    # ---------------------------------------------------------------- add
    def add(self, x, y, white_radius=None, yolk_radius=None, white_n_particles=None,
            yolk_n_particles=None):  # L:27-135
        white_particle_radius = mix(self._white_config["min_radius"], self._white_config["max_radius"], 0.5)
        yolk_particle_radius = mix(self._yolk_config["min_radius"], self._yolk_config["max_radius"], 0.5)
        if white_radius is None:
            white_radius = white_particle_radius * 15
        if yolk_radius is None:
            yolk_radius = white_radius * (10 / 50)
        if white_n_particles is None:
            white_n_particles = math.ceil(
                (math.pi * (white_radius * white_radius)) / (math.pi * (white_particle_radius * white_particle_radius)))
        if yolk_n_particles is None:
            yolk_n_particles = math.ceil(
                (math.pi * (yolk_radius * yolk_radius)) / (math.pi * (yolk_particle_radius * yolk_particle_radius)))
        if white_radius <= 0 or yolk_radius <= 0:
            raise ValueError("radius cannot be 0 or negative")
        if white_n_particles <= 1 or yolk_n_particles <= 1:
            raise ValueError("particle count cannot be 1 or negative")
        self._total_n_white_particles += white_n_particles
        self._total_n_yolk_particles += yolk_n_particles
        batch_id, batch = self._new_batch(x, y, white_radius, white_radius, white_n_particles,
                                          yolk_radius, yolk_radius, yolk_n_particles)
        self._batch_id_to_batch[batch_id] = batch
        self._n_batches += 1
        return batch_id

    def _new_batch(self, center_x, center_y, white_x_radius, white_y_radius, white_n_particles,
                   yolk_x_radius, yolk_y_radius, yolk_n_particles):  # L:881-1033
        batch = dict(white_particle_indices=[], yolk_particle_indices=[],
                     white_radius=max(white_x_radius, white_y_radius),
                     yolk_radius=max(yolk_x_radius, yolk_y_radius),
                     target_x=center_x, target_y=center_y)

        def fibonacci_spiral(i, n, x_radius, y_radius):  # L:907-918
            golden_ratio = (1 + math.sqrt(5)) / 2
            golden_angle = 2 * math.pi / (golden_ratio * golden_ratio)
            r = math.sqrt((i - 1) / n)
            theta = i * golden_angle
            return r * x_radius * math.cos(theta), r * y_radius * math.sin(theta)

        def get_mass(i, n):  # L:921-938
            variance = self._mass_distribution_variance

            def butterworth(t):
                u = variance * (t - 0.5)
                u2 = u * u
                return 1 / (1 + u2 * u2)  # (..)^4 by repeated squaring

            left = (i - 0.5) / n
            right = (i + 0.5) / n
            center = 0.5 * (left + right)
            half_width = 0.5 * (right - left)
            t1 = center - half_width / math.sqrt(3)
            t2 = center + half_width / math.sqrt(3)
            return 0.5 * (butterworth(t1) + butterworth(t2))

        def add_particle(array, config, x_radius, y_radius, particle_i, n_particles, batch_id):  # L:941-997
            dx, dy = fibonacci_spiral(particle_i, n_particles, x_radius, y_radius)
            x = center_x + dx
            y = center_y + dy
            t = get_mass(particle_i, n_particles)
            mass = mix(config["min_mass"], config["max_mass"], t)
            radius = mix(config["min_radius"], config["max_radius"], t)
            i = len(array)  # == #array + 1 because slot 0 is a dummy
            array.extend([0.0] * STRIDE)
            array[i + X] = x
            array[i + Y] = y
            array[i + Z] = 0
            array[i + VX] = 0
            array[i + VY] = 0
            array[i + PX] = x
            array[i + PY] = y
            array[i + RADIUS] = radius
            array[i + MASS_T] = t
            array[i + MASS] = mass
            array[i + INV_MASS] = 1 / mass
            array[i + CELL_X] = -INF
            array[i + CELL_Y] = -INF
            array[i + BATCH_ID] = batch_id
            array[i + R_] = array[i + G_] = array[i + B_] = array[i + A_] = 1
            array[i + LAST_X] = x
            array[i + LAST_Y] = y
            return i

        batch_id = self._current_batch_id
        self._current_batch_id += 1
        for i in range(1, white_n_particles + 1):
            batch["white_particle_indices"].append(
                add_particle(self._white_data, self._white_config, white_x_radius, white_y_radius, i,
                             white_n_particles, batch_id))
        for i in range(1, yolk_n_particles + 1):
            batch["yolk_particle_indices"].append(
                add_particle(self._yolk_data, self._yolk_config, yolk_x_radius, yolk_y_radius, i,
                             yolk_n_particles, batch_id))
        batch["n_white_particles"] = white_n_particles
        batch["n_yolk_particles"] = yolk_n_particles
        return batch_id, batch

    # ------------------------------------------------------------- remove
    def remove(self, batch_id):  # L:140-155
        batch = self._batch_id_to_batch.get(batch_id)
        if batch is None:
            return False
        del self._batch_id_to_batch[batch_id]
        self._n_batches -= 1
        self._total_n_white_particles -= batch["n_white_particles"]
        self._total_n_yolk_particles -= batch["n_yolk_particles"]
        self._remove(batch["white_particle_indices"], batch["yolk_particle_indices"])
        return True

    def _remove(self, white_indices, yolk_indices):  # L:1037-1106
        def remove_particles(indices, data, list_name):
            if not indices:
                return
            stride = STRIDE
            total_particles = (len(data) - 1) // stride
            remove = {}
            for base in indices:
                remove[(base - 1) // stride + 1] = True
            new_index = {}
            write = 0
            for read in range(1, total_particles + 1):
                if read not in remove:
                    write += 1
                    new_index[read] = write
            for read in range(1, total_particles + 1):
                write_i = new_index.get(read)
                if write_i and write_i != read:
                    src = (read - 1) * stride + 1
                    dst = (write_i - 1) * stride + 1
                    for o in range(stride):
                        data[dst + o] = data[src + o]
            del data[write * stride + 1:]
            for batch in self._batch_id_to_batch.values():
                lst = batch[list_name]
                out = []
                for old_base_index in lst:
                    old_particle_id = (old_base_index - 1) // stride + 1
                    new_particle_id = new_index.get(old_particle_id)
                    if new_particle_id:
                        out.append((new_particle_id - 1) * stride + 1)
                batch[list_name] = out

        remove_particles(white_indices, self._white_data, "white_particle_indices")
        remove_particles(yolk_indices, self._yolk_data, "yolk_particle_indices")

    # ------------------------------------------------------------ targets
    def set_target_position(self, batch_id, x, y):  # L:254-264
        batch = self._batch_id_to_batch.get(batch_id)
        if batch is None:
            return False
        batch["target_x"] = x
        batch["target_y"] = y
        return True

    def get_target_position(self, batch_id):  # L:268-278
        batch = self._batch_id_to_batch[batch_id]
        return batch["target_x"], batch["target_y"]

    def get_position(self, batch_id):  # L:281-295, L:1134-1148
        batch = self._batch_id_to_batch[batch_id]
        x, y = 0, 0
        for i in batch["white_particle_indices"]:
            x = x + self._white_data[i + X]
            y = y + self._white_data[i + Y]
        for i in batch["yolk_particle_indices"]:
            x = x + self._yolk_data[i + X]
            y = y + self._yolk_data[i + Y]
        n = batch["n_white_particles"] + batch["n_yolk_particles"]
        return x / n, y / n

    # ------------------------------------------------------------- update
    def update(self, delta, step_delta=None, n_substeps=None, n_collision_steps=None):  # L:168-222
        if step_delta is None:
            step_delta = 1 / 60
        if n_substeps is None:
            n_substeps = 2
        if n_collision_steps is None:
            n_collision_steps = 3
        n_substeps = math.ceil(n_substeps)
        n_collision_steps = math.ceil(n_collision_steps)
        if step_delta < 0 or step_delta != step_delta:
            raise ValueError("`step_delta` is not a number > 0")
        if n_substeps < 1:
            raise ValueError("`n_substeps` is not a number > 0")
        if n_collision_steps < 1:
            raise ValueError("`n_collision_steps` is not a number > 0")
        self._elapsed = self._elapsed + delta
        step = step_delta
        n_steps = 0
        max_n_steps = max(4, 4 * math.ceil((1 / 60) / step_delta))
        while self._elapsed >= step:
            self._step(step, n_substeps, n_collision_steps)
            self._elapsed = self._elapsed - step
            n_steps = n_steps + 1
            if n_steps > max_n_steps:
                self._elapsed = 0
                break
        self._interpolation_alpha = clamp(self._elapsed / step, 0, 1)
        return n_steps

    # --------------------------------------------------------------- step
    @staticmethod
    def _strength_to_compliance(strength, sub_step_delta):  # L:1337-1341
        alpha = 1 - clamp(strength, 0, 1)
        return alpha / (sub_step_delta * sub_step_delta)

    @staticmethod
    def _create_environment(current_env):  # L:1344-1390
        if current_env is None:
            return dict(particles=None, collided={}, spatial_hash={}, batch_id_to_follow_x={},
                        batch_id_to_follow_y={}, batch_id_to_radius={}, damping=1, min_x=INF, min_y=INF,
                        max_x=-INF, max_y=-INF, n_particles=0, centroid_x=0, centroid_y=0)
        env = current_env
        env["spatial_hash"].clear()
        env["collided"].clear()
        env["batch_id_to_follow_x"].clear()
        env["batch_id_to_follow_y"].clear()
        env["min_x"] = INF
        env["min_y"] = INF
        env["max_x"] = -INF
        env["max_y"] = -INF
        env["centroid_x"] = 0
        env["centroid_y"] = 0
        return env

    @staticmethod
    def _pre_solve(particles, n_particles, damping, delta, should_update_mass, min_mass, max_mass,
                   should_update_radius, min_radius, max_radius):  # L:1393-1432
        for particle_i in range(1, n_particles + 1):
            i = offset(particle_i)
            x, y = particles[i + X], particles[i + Y]
            particles[i + PX] = x
            particles[i + PY] = y
            velocity_x = particles[i + VX] * damping
            velocity_y = particles[i + VY] * damping
            particles[i + VX] = velocity_x
            particles[i + VY] = velocity_y
            particles[i + X] = x + delta * velocity_x
            particles[i + Y] = y + delta * velocity_y
            mass_t = particles[i + MASS_T]
            if should_update_mass:
                mass = mix(min_mass, max_mass, mass_t)
                particles[i + MASS] = mass
                particles[i + INV_MASS] = 1 / mass
            if should_update_radius:
                particles[i + RADIUS] = mix(min_radius, max_radius, mass_t)

    @staticmethod
    def _solve_follow_constraint(particles, n_particles, batch_id_to_radius, batch_id_to_follow_x,
                                 batch_id_to_follow_y, compliance):  # L:1435-1471
        for particle_i in range(1, n_particles + 1):
            i = offset(particle_i)
            batch_id = particles[i + BATCH_ID]
            follow_x = batch_id_to_follow_x[batch_id]
            follow_y = batch_id_to_follow_y[batch_id]
            x, y = particles[i + X], particles[i + Y]
            current_distance = distance(x, y, follow_x, follow_y)
            target_distance = 2 * batch_id_to_radius[batch_id]
            inverse_mass = particles[i + INV_MASS]
            if inverse_mass > EPS and current_distance > target_distance:
                dx, dy = normalize(follow_x - x, follow_y - y)
                constraint_violation = current_distance - target_distance
                delta_lambda = constraint_violation / (inverse_mass + compliance)
                x_correction = dx * delta_lambda * inverse_mass
                y_correction = dy * delta_lambda * inverse_mass
                particles[i + X] = particles[i + X] + x_correction
                particles[i + Y] = particles[i + Y] + y_correction

    @staticmethod
    def _rebuild_spatial_hash(particles, n_particles, spatial_hash, spatial_hash_cell_radius):  # L:1486-1511
        for particle_i in range(1, n_particles + 1):
            i = offset(particle_i)
            cell_x = math.floor(particles[i + X] / spatial_hash_cell_radius)
            cell_y = math.floor(particles[i + Y] / spatial_hash_cell_radius)
            particles[i + CELL_X] = cell_x
            particles[i + CELL_Y] = cell_y
            h = xy_to_hash(cell_x, cell_y)
            entry = spatial_hash.get(h)
            if entry is None:
                entry = []
                spatial_hash[h] = entry
            entry.append(particle_i)

    @staticmethod
    def _enforce_distance(ax, ay, bx, by, inverse_mass_a, inverse_mass_b, target_distance, compliance):  # L:1514-1545
        dx = bx - ax
        dy = by - ay
        current_distance = magnitude(dx, dy)
        dx, dy = normalize(dx, dy)
        constraint_violation = current_distance - target_distance
        mass_sum = inverse_mass_a + inverse_mass_b
        divisor = mass_sum + compliance
        if divisor < EPS:
            return 0, 0, 0, 0
        correction = -constraint_violation / divisor
        max_correction = abs(constraint_violation)
        correction = clamp(correction, -max_correction, max_correction)
        return (-dx * correction * inverse_mass_a, -dy * correction * inverse_mass_a,
                dx * correction * inverse_mass_b, dy * correction * inverse_mass_b)

    def _solve_collision(self, particles, n_particles, spatial_hash, collided, collision_overlap_factor,
                         collision_compliance, cohesion_interaction_distance_factor, cohesion_compliance,
                         max_n_collisions, visit_log=None):  # L:1548-1666
        n_collided = 0
        for self_particle_i in range(1, n_particles + 1):
            self_i = offset(self_particle_i)
            self_inverse_mass = particles[self_i + INV_MASS]
            self_radius = particles[self_i + RADIUS]
            self_batch_id = particles[self_i + BATCH_ID]
            cell_x = particles[self_i + CELL_X]
            cell_y = particles[self_i + CELL_Y]
            for x_offset in (-1, 0, 1):
                for y_offset in (-1, 0, 1):
                    entry = spatial_hash.get(xy_to_hash(cell_x + x_offset, cell_y + y_offset))
                    if entry is None:
                        continue
                    for other_particle_i in entry:
                        if self_particle_i == other_particle_i:
                            continue
                        pair_hash = xy_to_hash(min(self_particle_i, other_particle_i),
                                               max(self_particle_i, other_particle_i))
                        if collided.get(pair_hash) is True:
                            continue
                        collided[pair_hash] = True
                        other_i = offset(other_particle_i)
                        other_inverse_mass = particles[other_i + INV_MASS]
                        other_radius = particles[other_i + RADIUS]
                        other_batch_id = particles[other_i + BATCH_ID]
                        if self_inverse_mass + other_inverse_mass < EPS:
                            continue
                        # cohesion, L:1603-1630
                        self_x, self_y = particles[self_i + X], particles[self_i + Y]
                        other_x, other_y = particles[other_i + X], particles[other_i + Y]
                        if self_batch_id == other_batch_id:
                            interaction_distance = 0
                        else:
                            interaction_distance = cohesion_interaction_distance_factor * (self_radius + other_radius)
                        if (self_batch_id == other_batch_id and
                                squared_distance(self_x, self_y, other_x, other_y)
                                <= interaction_distance * interaction_distance):
                            c = self._enforce_distance(self_x, self_y, other_x, other_y, self_inverse_mass,
                                                       other_inverse_mass, interaction_distance, cohesion_compliance)
                            particles[self_i + X] = self_x + c[0]
                            particles[self_i + Y] = self_y + c[1]
                            particles[other_i + X] = other_x + c[2]
                            particles[other_i + Y] = other_y + c[3]
                        # collision, L:1632-1654
                        min_distance = collision_overlap_factor * (self_radius + other_radius)
                        self_x, self_y = particles[self_i + X], particles[self_i + Y]
                        other_x, other_y = particles[other_i + X], particles[other_i + Y]
                        dist = squared_distance(self_x, self_y, other_x, other_y)
                        if dist <= min_distance * min_distance:
                            c = self._enforce_distance(self_x, self_y, other_x, other_y, self_inverse_mass,
                                                       other_inverse_mass, min_distance, collision_compliance)
                            particles[self_i + X] = self_x + c[0]
                            particles[self_i + Y] = self_y + c[1]
                            particles[other_i + X] = other_x + c[2]
                            particles[other_i + Y] = other_y + c[3]
                        n_collided = n_collided + 1
                        if visit_log is not None:
                            visit_log.append((self_particle_i, other_particle_i))
                        if n_collided >= max_n_collisions:
                            return n_collided, True
        return n_collided, False

    @staticmethod
    def _post_solve(particles, n_particles, delta):  # L:1669-1718
        min_x, min_y = INF, INF
        max_x, max_y = -INF, -INF
        centroid_x, centroid_y = 0, 0
        max_velocity = 0
        max_radius = 0
        for particle_i in range(1, n_particles + 1):
            i = offset(particle_i)
            x = particles[i + X]
            y = particles[i + Y]
            velocity_x = (x - particles[i + PX]) / delta
            velocity_y = (y - particles[i + PY]) / delta
            particles[i + VX] = velocity_x
            particles[i + VY] = velocity_y
            velocity_magnitude = magnitude(velocity_x, velocity_y)
            if velocity_magnitude > max_velocity:
                max_velocity = velocity_magnitude
            centroid_x = centroid_x + x
            centroid_y = centroid_y + y
            r = particles[i + RADIUS]
            if r > max_radius:
                max_radius = r
            min_x = min(min_x, x - r)
            min_y = min(min_y, y - r)
            max_x = max(max_x, x + r)
            max_y = max(max_y, y + r)
        if n_particles > 0:
            centroid_x = centroid_x / n_particles
            centroid_y = centroid_y / n_particles
        return min_x, min_y, max_x, max_y, centroid_x, centroid_y, max_radius, max_velocity

    def _step(self, delta, n_sub_steps, n_collision_steps, visit_logs=None):  # L:1722-1989
        sub_delta = max(delta / n_sub_steps, EPS)

        def update_environment(old_env, config, particles, n_particles):  # L:1726-1774
            env = self._create_environment(old_env)
            env["particles"] = particles
            env["n_particles"] = n_particles
            if old_env is not None:
                env["should_update_mass"] = (config["min_mass"] != old_env["min_mass"]
                                             or config["max_mass"] != old_env["max_mass"])
                env["should_update_radius"] = (config["min_radius"] != old_env["min_radius"]
                                               or config["max_radius"] != old_env["max_radius"])
            else:
                env["should_update_mass"] = True
                env["should_update_radius"] = True
            env["min_mass"] = config["min_mass"]
            env["max_mass"] = config["max_mass"]
            env["min_radius"] = config["min_radius"]
            env["max_radius"] = config["max_radius"]
            env["max_n_collisions"] = self._max_collision_fraction * (env["n_particles"] * env["n_particles"])
            max_factor = max(config["collision_overlap_factor"], config["cohesion_interaction_distance_factor"])
            env["spatial_hash_cell_radius"] = max(1, config["max_radius"] * max_factor)
            for batch_id, batch in self._batch_id_to_batch.items():
                env["batch_id_to_follow_x"][batch_id] = batch["target_x"]
                env["batch_id_to_follow_y"][batch_id] = batch["target_y"]
            env["damping"] = 1 - clamp(config["damping"], 0, 1)
            env["follow_compliance"] = self._strength_to_compliance(config["follow_strength"], sub_delta)
            env["collision_compliance"] = self._strength_to_compliance(config["collision_strength"], sub_delta)
            env["cohesion_compliance"] = self._strength_to_compliance(config["cohesion_strength"], sub_delta)
            return env

        white_config = self._white_config
        white_env = update_environment(self._last_white_env, white_config, self._white_data,
                                       self._total_n_white_particles)
        yolk_config = self._yolk_config
        yolk_env = update_environment(self._last_yolk_env, yolk_config, self._yolk_data,
                                      self._total_n_yolk_particles)

        for batch_id, batch in self._batch_id_to_batch.items():  # L:1789-1792
            white_env["batch_id_to_radius"][batch_id] = math.sqrt(batch["white_radius"])
            yolk_env["batch_id_to_radius"][batch_id] = math.sqrt(batch["yolk_radius"])

        def update_last_positions(env):  # L:1795-1815
            particles = env["particles"]
            sum_x, sum_y = 0, 0
            for particle_i in range(1, env["n_particles"] + 1):
                i = offset(particle_i)
                x = particles[i + X]
                y = particles[i + Y]
                particles[i + LAST_X] = x
                particles[i + LAST_Y] = y
                sum_x = sum_x + x
                sum_y = sum_y + y
            if env["n_particles"] > 0:
                env["last_centroid_x"] = sum_x / env["n_particles"]
                env["last_centroid_y"] = sum_y / env["n_particles"]
            else:
                env["last_centroid_x"] = 0
                env["last_centroid_y"] = 0

        update_last_positions(white_env)
        update_last_positions(yolk_env)

        self.pass_log = []
        for sub_step_i in range(1, n_sub_steps + 1):
            for env in (white_env, yolk_env):
                self._pre_solve(env["particles"], env["n_particles"], env["damping"], sub_delta,
                                env["should_update_mass"], env["min_mass"], env["max_mass"],
                                env["should_update_radius"], env["min_radius"], env["max_radius"])
            for env in (white_env, yolk_env):
                self._solve_follow_constraint(env["particles"], env["n_particles"], env["batch_id_to_radius"],
                                              env["batch_id_to_follow_x"], env["batch_id_to_follow_y"],
                                              env["follow_compliance"])
            for collision_i in range(1, n_collision_steps + 1):
                for env in (white_env, yolk_env):
                    self._rebuild_spatial_hash(env["particles"], env["n_particles"], env["spatial_hash"],
                                               env["spatial_hash_cell_radius"])
                for which, (env, config) in enumerate(((white_env, white_config), (yolk_env, yolk_config))):
                    log = None
                    if visit_logs is not None:
                        log = []
                        visit_logs.append((which, sub_step_i - 1, collision_i - 1, log))
                    n, cut = self._solve_collision(env["particles"], env["n_particles"], env["spatial_hash"],
                                                   env["collided"], config["collision_overlap_factor"],
                                                   env["collision_compliance"],
                                                   config["cohesion_interaction_distance_factor"],
                                                   env["cohesion_compliance"], env["max_n_collisions"], log)
                    self.pass_log.append((which, sub_step_i - 1, collision_i - 1, n, cut))
                if collision_i < n_collision_steps:  # L:1905-1912
                    white_env["spatial_hash"].clear()
                    white_env["collided"].clear()
                    yolk_env["spatial_hash"].clear()
                    yolk_env["collided"].clear()
            for env in (white_env, yolk_env):
                (env["min_x"], env["min_y"], env["max_x"], env["max_y"], env["centroid_x"], env["centroid_y"],
                 env["max_radius"], env["max_velocity"]) = self._post_solve(env["particles"], env["n_particles"],
                                                                            sub_delta)
        self._last_white_env = white_env
        self._last_yolk_env = yolk_env
        self.n_steps += 1

    # ------------------------------------------------------------ readout
    def field(self, which, off):
        data = self._white_data if which == 0 else self._yolk_data
        n = self._total_n_white_particles if which == 0 else self._total_n_yolk_particles
        return [data[offset(p) + off] for p in range(1, n + 1)]
