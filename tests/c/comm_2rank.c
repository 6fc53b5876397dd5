/* Plain-C consumer of the multi-GPU part of include/r3d.h: forks one process per GPU (2 by default), rank 0 makes the
 * RCCL id and hands it over a pipe, every rank fuses its own block of frames and the ranks assemble the world cloud
 * with r3d_allgather_xyz (unequal shards: rank 0 holds one frame more).  No torch, no Python.
 * Exit codes: 0 ok, 77 fewer GPUs than ranks (RCCL needs one GPU per rank), 1 failure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include "r3d.h"

#define H 24
#define W 32
#define CHECK(x)                                                              \
  do {                                                                        \
    int rc_ = (x);                                                            \
    if (rc_ != R3D_OK) {                                                      \
      fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #x, rc_, r3d_last_error()); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

static void make_job(int n_frames, unsigned char* depth, double* pose) {
  int f, k;
  for (k = 0; k < n_frames * H * W; ++k) depth[k] = (unsigned char)(1 + (k * 7 + k / 13) % 255);
  for (f = 0; f < n_frames; ++f) {
    double* p = pose + f * 12;
    for (k = 0; k < 12; ++k) p[k] = 0.0;
    p[0] = p[4] = p[8] = 1.0; /* identity rotation, per-frame translation */
    p[9] = f;
    p[10] = -2.0 * f;
    p[11] = 0.5 * f;
  }
}

static int run_rank(int rank, int world, const unsigned char* id, int n_frames, int algo, int n_gpus) {
  r3d_ctx* ctx = NULL;
  r3d_camera* cam = NULL;
  r3d_comm* comm = NULL;
  int64_t pts[16];
  int lo = 0, hi = 0, r, per = (n_frames + world - 1) / world;
  size_t n_all = (size_t)n_frames * H * W;
  unsigned char* depth = (unsigned char*)malloc(n_all);
  double* pose = (double*)malloc((size_t)n_frames * 12 * sizeof(double));
  float *got = (float*)malloc(n_all * 12), *want = (float*)malloc(n_all * 12);
  void *d_depth = NULL, *d_pose = NULL, *d_full = NULL;
  make_job(n_frames, depth, pose);
  for (r = 0; r < world; ++r) {
    int l = r * per < n_frames ? r * per : n_frames, h = l + per < n_frames ? l + per : n_frames;
    pts[r] = (int64_t)(h - l) * H * W;
    if (r == rank) {
      lo = l;
      hi = h;
    }
  }
  CHECK(r3d_ctx_create(rank % n_gpus, NULL, 0, &ctx)); /* rank % n_gpus only matters for the shared-GPU rehearsal */
  CHECK(r3d_camera_create(ctx, H, W, 600.391, 600.079, 320, 240, &cam));
  CHECK(r3d_comm_create(ctx, id, rank, world, &comm));
  CHECK(r3d_dev_alloc(ctx, n_all, &d_depth));
  CHECK(r3d_dev_alloc(ctx, (size_t)n_frames * 96, &d_pose));
  CHECK(r3d_dev_alloc(ctx, n_all * 12, &d_full));
  CHECK(r3d_memcpy_h2d(ctx, d_depth, depth, n_all));
  CHECK(r3d_memcpy_h2d(ctx, d_pose, pose, (size_t)n_frames * 96));
  /* single-GPU answer over all frames (every rank can compute it: the check) */
  CHECK(r3d_fuse_frames(ctx, cam, d_depth, R3D_DEPTH_U8, n_frames, 1.0, (const double*)d_pose, d_full, R3D_F32));
  CHECK(r3d_memcpy_d2h(ctx, want, d_full, n_all * 12));
  CHECK(r3d_memset(ctx, d_full, 0xff, n_all * 12));
  /* sharded: own frames straight into this rank's slot, then the exchange step */
  {
    char* slot = (char*)d_full + (size_t)lo * H * W * 12;
    CHECK(r3d_fuse_frames(ctx, cam, (char*)d_depth + (size_t)lo * H * W, R3D_DEPTH_U8, hi - lo, 1.0,
                          (const double*)d_pose + (size_t)lo * 12, slot, R3D_F32));
    CHECK(r3d_allgather_xyz(comm, slot, pts, R3D_F32, d_full, algo));
  }
  CHECK(r3d_memcpy_d2h(ctx, got, d_full, n_all * 12));
  CHECK(r3d_ctx_sync(ctx));
  if (memcmp(got, want, n_all * 12) != 0) {
    fprintf(stderr, "rank %d: gathered cloud differs from the single-GPU cloud\n", rank);
    return 1;
  }
  /* the other assembly: all-gather the rasters + pose rows of every rank, then ONE fused launch over all frames */
  {
    int64_t fpr[16];
    void *d_depth_all = NULL, *d_pose_all = NULL;
    for (r = 0; r < world; ++r) fpr[r] = pts[r] / (H * W);
    CHECK(r3d_dev_alloc(ctx, n_all, &d_depth_all));
    CHECK(r3d_dev_alloc(ctx, (size_t)n_frames * 96, &d_pose_all));
    CHECK(r3d_memset(ctx, d_full, 0xff, n_all * 12));
    CHECK(r3d_allgather_inputs(comm, (char*)d_depth + (size_t)lo * H * W, R3D_DEPTH_U8, fpr, H, W,
                               (const double*)d_pose + (size_t)lo * 12, d_depth_all, (double*)d_pose_all, algo));
    CHECK(r3d_fuse_frames(ctx, cam, d_depth_all, R3D_DEPTH_U8, n_frames, 1.0, (const double*)d_pose_all, d_full, R3D_F32));
    CHECK(r3d_memcpy_d2h(ctx, got, d_full, n_all * 12));
    CHECK(r3d_ctx_sync(ctx));
    if (memcmp(got, want, n_all * 12) != 0) {
      fprintf(stderr, "rank %d: cloud fused from gathered inputs differs from the single-GPU cloud\n", rank);
      return 1;
    }
    r3d_dev_free(ctx, d_depth_all);
    r3d_dev_free(ctx, d_pose_all);
  }
  /* all-reduce: every rank contributes rank+1 in 18 slots */
  {
    double h[18], *d_s = NULL;
    int k;
    for (k = 0; k < 18; ++k) h[k] = (rank + 1) * (k + 1);
    CHECK(r3d_dev_alloc(ctx, sizeof(h), (void**)&d_s));
    CHECK(r3d_memcpy_h2d(ctx, d_s, h, sizeof(h)));
    CHECK(r3d_comm_allreduce_sum_f64(comm, d_s, 18));
    CHECK(r3d_memcpy_d2h(ctx, h, d_s, sizeof(h)));
    CHECK(r3d_ctx_sync(ctx));
    for (k = 0; k < 18; ++k)
      if (h[k] != (k + 1) * (world * (world + 1) / 2)) {
        fprintf(stderr, "rank %d: all-reduce slot %d = %g\n", rank, k, h[k]);
        return 1;
      }
    r3d_dev_free(ctx, d_s);
  }
  printf("rank %d of %d: frames [%d,%d) fused, %lld points assembled, identical to the single-GPU cloud\n", rank, world, lo, hi,
         (long long)(n_all));
  r3d_comm_destroy(comm);
  r3d_camera_destroy(cam);
  r3d_dev_free(ctx, d_depth);
  r3d_dev_free(ctx, d_pose);
  r3d_dev_free(ctx, d_full);
  r3d_ctx_destroy(ctx);
  free(depth);
  free(pose);
  free(got);
  free(want);
  return 0;
}

int main(int argc, char** argv) {
  int world = argc > 1 ? atoi(argv[1]) : 2, n_frames = argc > 2 ? atoi(argv[2]) : 5;
  int algo = argc > 3 ? atoi(argv[3]) : R3D_GATHER_AUTO;
  const int share_gpu = getenv("R3D_SHARE_GPU") != NULL; /* rehearsal with a stand-in transport: ranks share GPU 0 */
  int n_gpus = 0, rank, status = 0, fail = 0;
  int pipes[16][2];
  pid_t pids[16];
  unsigned char id[R3D_COMM_ID_BYTES];
  printf("r3d version %d\n", r3d_version());
  fflush(stdout); /* before any fork: children inherit stdio buffers */
  if (world < 1 || world > 16) return 1;
  /* counting devices happens in a child: the parent must not initialise the GPU before it forks */
  {
    int pc[2];
    pid_t p;
    if (pipe(pc)) return 1;
    p = fork();
    if (p == 0) {
      int n = 0;
      if (r3d_device_count(&n) != R3D_OK) n = 0;
      if (write(pc[1], &n, sizeof(n)) != sizeof(n)) _exit(1);
      _exit(0);
    }
    if (read(pc[0], &n_gpus, sizeof(n_gpus)) != sizeof(n_gpus)) n_gpus = 0;
    waitpid(p, &status, 0);
  }
  if (n_gpus >= 1 && share_gpu) {
    printf("R3D_SHARE_GPU: %d ranks share %d GPU(s) (needs a transport that allows it, e.g. the test's mock)\n", world, n_gpus);
    fflush(stdout);
  } else if (n_gpus < world) {
    printf("%d GPU(s) visible, %d ranks wanted: RCCL needs one GPU per rank -- skipped\n", n_gpus, world);
    fflush(stdout);
    return 77;
  }
  for (rank = 0; rank < world; ++rank)
    if (pipe(pipes[rank])) return 1;
  for (rank = 0; rank < world; ++rank) {
    pids[rank] = fork();
    if (pids[rank] == 0) {
      int r;
      if (rank == 0) {
        if (r3d_comm_unique_id(id) != R3D_OK) {
          fprintf(stderr, "unique id: %s\n", r3d_last_error());
          _exit(1);
        }
        for (r = 1; r < world; ++r)
          if (write(pipes[r][1], id, sizeof(id)) != (ssize_t)sizeof(id)) _exit(1);
      } else if (read(pipes[rank][0], id, sizeof(id)) != (ssize_t)sizeof(id)) {
        _exit(1);
      }
      {
        int rc = run_rank(rank, world, id, n_frames, algo, n_gpus > 0 ? n_gpus : 1);
        fflush(NULL);
        _exit(rc);
      }
    }
  }
  for (rank = 0; rank < world; ++rank) {
    waitpid(pids[rank], &status, 0);
    if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) fail = 1;
  }
  return fail;
}
