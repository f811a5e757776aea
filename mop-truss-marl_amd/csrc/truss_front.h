// truss_front.h -- batched Pareto front + 2-D hypervolume (include/truss_mi355.h, truss_front).
// Shared argument checks; the gfx950 kernel lives in truss_hip.hip, a serial C++ restatement for the
// CPU test backend in tests/emu/truss_emu.cpp.
#pragma once
#include <cstdint>
#include <string>

static int tb_fail(int code, const std::string &msg);

static inline int tb_front_check(const truss_front_args_t *a) {
  if (!a || a->struct_size != sizeof(truss_front_args_t)) return tb_fail(TRUSS_EINVAL, "truss_front: bad args / struct_size");
  if (a->n_envs < 0 || a->max_points < 1 || a->max_points > TRUSS_FRONT_MAXP)
    return tb_fail(TRUSS_EINVAL, "truss_front: max_points must be 1..64");
  if (!a->points || !a->n_points) return tb_fail(TRUSS_EINVAL, "truss_front: points / n_points NULL");
  if ((a->flags & TRUSS_FRONT_TRUNCATE) && a->max_front < 2) return tb_fail(TRUSS_EINVAL, "truss_front: max_front < 2");
  return TRUSS_OK;
}
