/*
 * fenicssolver_amd.h — C-ABI of libfsamd.so, the MI355X (gfx950) replacement for
 * the part of FenicsSolver that the reference delegates to DOLFIN/FFC/PETSc.
 *
 * The reference has no FFI of its own: the seam is the Python methods
 *   SolverBase.solve_linear_problem(F, u, bcs)   FenicsSolver/SolverBase.py:592-613
 *   SolverBase.solve_amg(F, u, bcs)              FenicsSolver/SolverBase.py:643-672
 *   SolverBase.solve_nonlinear_problem(...)      FenicsSolver/SolverBase.py:615-626
 * which call dolfin.assemble / assemble_system / DirichletBC.apply /
 * LinearVariationalSolver / PETScKrylovSolver.  Each entry point below names the
 * dolfin call (and the reference line that makes it) that it stands in for; the
 * ctypes binding a maintainer would add is fenicssolver_amd/_lib.py (see
 * INTEGRATION.md).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no exceptions cross the boundary.
 *   - every function returns 0 (FS_OK) or a negative FS_ERR_* code; the message is
 *     available from fs_last_error() (thread-local).
 *   - handles are opaque and own device (HBM) memory; host arrays passed in are
 *     borrowed for the duration of the call only (C-contiguous, fp64 / int32 / int64).
 *   - calls are synchronous at return unless stated otherwise.
 *   - one process drives one GPU; N-GPU runs are N processes joined through
 *     fs_comm_init (RCCL).  All field arithmetic is fp64, connectivity is int32.
 *   - threading: one thread at a time per process, as for the PETSc objects of one communicator.  All work is
 *     ordered on the library's single HIP stream, the device block cache and the Krylov / AMG / saddle-point work
 *     spaces (and fs_krylov_history) belong to the process.  The three solve entry points (fs_krylov_solve, fs_amg_solve,
 *     fs_saddle_solve) take a process-wide lock and the block cache its own, so concurrent calls from several threads are
 *     serialised, not undefined; assembly calls on DIFFERENT handles may overlap on the host but still share the stream.
 *     Concurrency comes from running one process per GPU.
 */
#ifndef FENICSSOLVER_AMD_H
#define FENICSSOLVER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_OK 0
#define FS_ERR_INVALID (-1)   /* bad argument / handle */
#define FS_ERR_HIP (-2)       /* HIP runtime error (message has the hipError string) */
#define FS_ERR_NO_DEVICE (-3) /* no gfx950 device visible: the library never falls back to the CPU */
#define FS_ERR_UNSUPPORTED (-4)
#define FS_ERR_COMM (-5)      /* RCCL error */
#define FS_ERR_NUMERIC (-6)   /* Krylov breakdown (non-SPD operator, NaN) */
#define FS_ERR_P2P_TIMEOUT (-7) /* a wait of the peer-to-peer halo exchange timed out (FS_P2P_TIMEOUT_MS): the transport failed, not the solve;
                                 * the host side turns the exchange of the space off on every rank and solves again over RCCL */

typedef struct fs_mesh_s* fs_mesh_t;
typedef struct fs_space_s* fs_space_t;
typedef struct fs_matrix_s* fs_matrix_t;
typedef struct fs_vector_s* fs_vector_t;

/* ---- runtime ---------------------------------------------------------------- */

/* Select the HIP device this process drives (dolfin has no analogue; MPI rank ->
 * device mapping replaces `mpirun`, SolverBase.py:102-118).  Also loads the library's code objects onto the device (about 70 ms,
 * half of it the set-up kernels' object with its rocPRIM sorts) so that the first mesh, space and solve of the process do not pay
 * for it; environment FS_PRELOAD=0 leaves the loading to the first launch out of each object.
 * Other environment switches read once per process (A/B and fall-backs, not needed for normal use): FS_PATTERN_BY_ROWS=0 (CG1
 * sparsity patterns through sorted node-pair keys instead of row by row), FS_DICT_REUSE=0 (row classes found from scratch at every
 * solve instead of compared with the kept table), FS_CG_FUSED / FS_CG_FUSED_MAX_ROWS / FS_CG_FUSED_P2P (the one-launch CG
 * iteration: option "cg_fused"), FS_SPMV_DICT=0 (no row-dictionary product: option "row_dictionary"). */
int fs_init(int device_id);
int fs_device_count(int* count);
int fs_device_synchronize(void);
/* Enqueues an empty kernel named k_profile_marker on the library's stream (no dolfin analogue; the reference's only timing is the
 * wall clock around solve(), SolverBase.py:514-525).  A traced command (rocprofv3 --kernel-trace / --pmc) that runs several problem
 * sizes or kernel variants back to back calls it between them; the n-th marker opens phase n of tools/summarize_profiles.py. */
int fs_profile_marker(int phase);
/* Device blocks released by the library are cached for re-use (hipFree synchronises the device, a time loop that
 * re-assembles its operators must not pay it every step): fs_memory_info reports the bytes in use / idle in the
 * cache, fs_memory_trim returns the idle ones to the driver.  FS_POOL_MAX_MB (environment, default 16384) caps the
 * idle bytes; 0 turns the cache off. */
int fs_memory_info(int64_t* live_bytes, int64_t* cached_bytes);
int fs_memory_trim(void);
const char* fs_last_error(void);
const char* fs_version(void);
/* Tunables: "spmv_blocks" (persistent SpMV grid, multiple of 8), "spmv_unroll"
 * (2/4/8/16 row entries in flight per lane), "spmv_unroll4" (1/2/4: the same for 4 x 4 block rows), "cg_batch" (iterations per host poll), "lattice_order" (-1 / 0 / 1: scalar CG2 operators on uniform boxes solved in
 * the lattice order of the half grid, fs_krylov_stats.lattice_order - automatic = from 270 000 rows on, where the product is then the
 * marching-window product k_lat_march or the tile product k_lattice_spmv / never / wherever the order exists), "lattice_march" (1 / 0, round 6: k_lat_march where
 * its tables can be built - every class inside its parity's compile-time stencil, every mesh line of one class per parity away from its ends - / the tile product; the same bits),
 * "lattice_check" (0 / 1: every solve in lattice order first compares that product with the work-item product on a
 * vector of pseudo-random numbers, every row, bit for bit - a difference fails the solve with FS_ERR_NUMERIC), "cg_mirror" (0 / 1: the one-launch iteration reports its progress
 * through pinned host memory and the host keeps "cg_ahead" to "cg_ahead" + "cg_sub" launches enqueued, instead of batches of
 * "cg_batch" with the status word copied back behind each),
 * "cg_fuse_sums" (0/1: sum the dot partials inside the update kernel on one GPU),
 * "update_blocks" (grid of the fused vector-update kernel), "cg_graph" (-1 / 0 / 1: CG batches as hipGraphs by size /
 * never / always), "cg_fused" (-1 / 0 / 1: ONE launch per CG iteration on row-dictionary operators - up to 3 M rows /
 * never / wherever it applies; fs_krylov_stats.fused_iteration), "row_dictionary" (0 / 1: allow the row-dictionary form of the product, fs_krylov_stats.row_classes),
 * "box_snap" (0 / 1: box meshes snap their edge vectors to the grid spacing so that equal stencils are equal bit for bit),
 * "box_assembly" (1 / 0: scalar CG1 operators on fs_mesh_create_box meshes are assembled from the reference rows of the six cell types instead of
 * per-incidence geometry - the same bits), "box_spmv" (1 / 0) and "box_min_rows" (default 1 500 000): the marching-window product k_box_spmv,
 * "box_iter" (0 / 1, default 0: opt-in) and "box_iter_min_rows" (default 400 000): the one-launch CG iteration of P1 box operators in marching-window
 * form (k_box_cg_iter; iterates equal to the other forms to rounding, not bit for bit; measured slower than k_dict_cg_iter up to 3 M rows),
 * "amg_coarse_fp32" (1 / 0, default 1: hierarchies built from now on keep the 6 x 6-block coarse operators and the transfer operators
 * the V-cycle streams rounded to fp32 - vectors, accumulation, the fine level and the CG outside stay fp64; fs_amg_level_get keeps
 * returning the fp64 operators; FS_AMG_FP32=0 in the environment is the same switch). */
int fs_set_option(const char* name, double value);
/* Name, CU count and HBM bytes of the selected device. */
int fs_device_info(char* name, int name_len, int* compute_units, int64_t* hbm_bytes);

/* ---- mesh  (dolfin.Mesh / BoxMesh, SolverBase.py:203-258) -------------------- */

/* Upload a tetrahedral mesh (gdim 3, 4 vertices per cell) or a triangular mesh (gdim 2, 3 vertices per
 * cell: scalar CG1 spaces, one GPU; facet lists of the boundary calls are then [n_facets][2] edge vertex pairs).
 * xyz[nv][gdim], cells[nc][verts_per_cell] vertex indices.  The first n_owned vertices are the rows this process owns
 * (n_owned == nv on one GPU); the remainder are ghost vertices whose values
 * arrive by halo exchange. */
int fs_mesh_create(int gdim, int64_t nv, const double* xyz, int64_t nc, const int32_t* cells,
                   int verts_per_cell, int64_t n_owned, fs_mesh_t* out);

/* Generate on the device the slab of dolfin.BoxMesh(p0, p1, nx, ny, nz) whose
 * owned vertex planes are [zplane_begin, zplane_end) (0 .. nz+1 for the whole
 * mesh).  Vertex order x-fastest, six tets per hex around the v0-v7 diagonal,
 * cells iz->iy->ix, each cell's vertices ascending by global index (SURVEY.md
 * section 8d / Appendix D-8).  Local numbering: owned planes first, then the
 * lower ghost plane, then the upper ghost plane. */
int fs_mesh_create_box(int64_t nx, int64_t ny, int64_t nz, const double p0[3], const double p1[3],
                       int64_t zplane_begin, int64_t zplane_end, fs_mesh_t* out);

/* Global vertex ids of an uploaded mesh part (identity by default).  Needed for CG2 spaces on a decomposed mesh:
 * an edge node is owned by the rank owning its endpoint of smaller global id. */
int fs_mesh_set_global_ids(fs_mesh_t mesh, const int64_t* global_ids);
int fs_mesh_info(fs_mesh_t mesh, int64_t* nv, int64_t* nc, int64_t* n_owned);
/* Copy back to host (any pointer may be NULL): xyz[nv][3], cells[nc][4],
 * global vertex ids[nv] (identity for uploaded meshes). */
int fs_mesh_get(fs_mesh_t mesh, double* xyz, int32_t* cells, int64_t* global_ids);
int fs_mesh_destroy(fs_mesh_t mesh);
/* Locality order of a mesh that arrives in file order (DOLFIN reorders the dofs of every FunctionSpace when it builds
 * the dofmap, SolverBase.py:260-275): vertex_order[k] = the vertex to upload k-th (Morton curve of the coordinates),
 * cell_order[k] = the cell to upload k-th (by the smallest new index of its vertices, file order among equals).  Computed
 * on the device; host arrays in, host arrays out; nothing is created.  The caller uploads xyz[vertex_order] and the
 * re-indexed cells[cell_order] through fs_mesh_create and keeps the permutation to translate indices and results. */
int fs_mesh_locality_order(int gdim, int64_t nv, const double* xyz, int64_t nc, const int32_t* cells, int verts_per_cell,
                           int32_t* vertex_order, int32_t* cell_order);
/* fs_mesh_locality_order + fs_mesh_create in one: the tetrahedral mesh of a FILE (Mesh(filename), SolverBase.py:203-258) uploaded
 * once, ordered on the device, and built there in that order - coordinates gathered, cells gathered and renamed, the vertices'
 * global ids = their numbers in the file.  vertex_order[k] / cell_order[c] = file number of new vertex k / new cell c, for the
 * host's maps between file numbering (what the API speaks) and device numbering. */
int fs_mesh_create_renumbered(int64_t nv, const double* xyz, int64_t nc, const int32_t* cells, int32_t* vertex_order,
                              int32_t* cell_order, fs_mesh_t* out);

/* ---- function space + sparsity (dolfin.FunctionSpace, SolverBase.py:260-275;
 *      the sparsity pattern DOLFIN builds inside the first assemble()) --------- */

#define FS_FAMILY_CG 0
/* degree 1 with ncomp = 1 (scalar) or 3 (vector, node-interleaved dofs as DOLFIN's
 * VectorFunctionSpace lays them out; 2 on triangular meshes: plane-strain elasticity); triangular meshes also carry
 * degree-2 spaces, scalar or 2-vector (3 vertices + 3 edge nodes per cell, UFC edge i opposite vertex i); degree 2 with ncomp = 1 (scalar) or 4 (Taylor-Hood block u_x,u_y,u_z,p):
 * nodes = the vertices, then one node per edge, edges numbered lexicographically by their ascending vertex
 * pair - grouped by index difference first on structured meshes.  With ghost vertices
 * (n_owned < nv) the nodes are [owned vertices | owned edges | ghost vertices | ghost edges]; an edge belongs to
 * the rank owning its endpoint of smaller global id (fs_mesh_set_global_ids). */
int fs_space_create(fs_mesh_t mesh, int family, int degree, int ncomp, fs_space_t* out);
/* The same space with additional node couplings in the sparsity pattern (node_pairs [n_pairs][2], both directions are
 * added): what DOLFIN's pattern builder does for a form with interior-facet (dS) integrals, where the two vertices
 * opposite a facet couple although they share no cell (ScalarTransportSolver.py:312-315), and for a periodic space, whose
 * folded system (fs_matrix_tie_nodes) couples every master with the neighbours of its slave.  CG1 and CG2. */
int fs_space_create_coupled(fs_mesh_t mesh, int family, int degree, int ncomp, int64_t n_pairs, const int32_t* node_pairs,
                            fs_space_t* out);
int fs_space_info(fs_space_t space, int64_t* n_dofs_local, int64_t* n_dofs_owned, int64_t* nnz,
                  int64_t* sell_entries);
/* A += coefficient * avg(h)^2 jump(grad u, n) jump(grad v, n) dS over the listed interior facets, h = 2 circumradius
 * (the 'IP' stabilisation of ScalarTransportSolver.py:312-315, coefficient = alpha * capacity).  facet_cells
 * [n_facets][2]: the two local cells of every facet.  Scalar CG1 on tetrahedra, space from fs_space_create_coupled. */
int fs_assemble_interior_penalty(fs_matrix_t A, int64_t n_facets, const int32_t* facet_cells, double coefficient);

/* Storage form chosen per 64-row slice: SELL (values + 4-B columns) or DIA (values only, the
 * 64 rows share one list of column offsets).  spmv_bytes = matrix bytes one SpMV streams. */
int fs_space_format_info(fs_space_t space, int64_t* n_slices, int64_t* n_dia_slices, int64_t* spmv_matrix_bytes);
/* CG2: the edge table [n_edges][2] (edge node e is dof n_vertices + e); n_edges = 0 for CG1. */
int fs_space_get_edges(fs_space_t space, int64_t* n_edges, int32_t* edges);
int fs_space_destroy(fs_space_t space);

/* ---- vectors (dolfin.Function.vector(), PETScVector) ------------------------ */

int fs_vector_create(int64_t n, fs_vector_t* out);
int fs_vector_size(fs_vector_t v, int64_t* n);
int fs_vector_set(fs_vector_t v, const double* host, int64_t n);
int fs_vector_get(fs_vector_t v, double* host, int64_t n);
int fs_vector_fill(fs_vector_t v, double value);      /* enqueued on the library's stream: returns without waiting for the kernel */
int fs_vector_axpy(fs_vector_t y, double a, fs_vector_t x); /* y += a x */
int fs_vector_copy(fs_vector_t dst, fs_vector_t src, int64_t n); /* first n entries, device to device (Function.assign) */
/* v[idx[k]] += vals[k] (repeated indices accumulate): dolfin.PointSource.apply(b), SolverBase.py:597-601 */
int fs_vector_add_entries(fs_vector_t v, int64_t n, const int32_t* idx, const double* vals);
int fs_vector_dot(fs_vector_t x, fs_vector_t y, double* result); /* local (un-reduced) dot */
int fs_vector_destroy(fs_vector_t v);

/* ---- matrices (PETSc AIJ behind dolfin.assemble, SolverBase.py:595, 608-612, 644) */

/* A matrix on the space's sparsity pattern, values zero.  Rows = owned dofs,
 * columns = local (owned + ghost) dofs. */
int fs_matrix_create(fs_space_t space, fs_matrix_t* out);
int fs_matrix_info(fs_matrix_t A, int64_t* n_rows, int64_t* n_cols, int64_t* nnz);
int fs_matrix_zero(fs_matrix_t A);
/* Y += a X (same space). Used for theta-scheme operators M/dt + theta K
 * (ScalarTransportSolver.py:287-293). */
int fs_matrix_axpy(fs_matrix_t Y, double a, fs_matrix_t X);
/* dst = src (same space), enqueued on the library's stream: a time loop with constant coefficients keeps its
 * unconstrained operator and starts every step from a copy (DOLFIN re-assembles, SolverBase.py:592-602). */
int fs_matrix_copy(fs_matrix_t dst, fs_matrix_t src);
/* Periodic constraints (FunctionSpace(..., constrained_domain=pb), SolverBase.py:260-275): node slaves[i] takes the
 * value of node masters[i] (chains resolved by the caller, a master is never a slave).  DOLFIN removes the slave dofs;
 * here the assembled system is folded in place - A <- P^T A P with a unit diagonal on the slave rows, b <- P^T b with 0
 * on the slaves (b may be NULL) - and after the solve fs_vector_assign_entries copies the masters' values to the
 * slaves.  The pattern must hold (master, j) and (master, fold(j)) for every neighbour j of a slave: create the space
 * with fs_space_create_coupled.  Apply before fs_apply_dirichlet.  Decomposed spaces: local node numbers, ghosts included -
 * the columns of every local slave fold onto its master in the rows this rank owns, the row of a slave folds on the rank that
 * owns it, which must own its master too. */
int fs_matrix_tie_nodes(fs_matrix_t A, fs_vector_t b, int64_t n_pairs, const int32_t* slaves, const int32_t* masters);
/* v[dst_nodes[i]*block + c] = v[src_nodes[i]*block + c], c < block. */
int fs_vector_assign_entries(fs_vector_t v, int64_t n, const int32_t* dst_nodes, const int32_t* src_nodes, int block);
/* Export as sorted-column CSR (any pointer may be NULL): rowptr[n_rows+1],
 * colidx[nnz], vals[nnz]. */
int fs_matrix_get_csr(fs_matrix_t A, int32_t* rowptr, int32_t* colidx, double* vals);
int fs_matrix_destroy(fs_matrix_t A);

/* Coefficient of a volume term: constant, one value per cell (DG0, e.g. a
 * per-subdomain material, SolverBase.py:331-332) or a constant 3x3 tensor
 * (as_matrix, SolverBase.py:327-330). */
#define FS_COEF_NONE 0
#define FS_COEF_CONST 1
#define FS_COEF_CELL 2
#define FS_COEF_TENSOR 3
#define FS_COEF_NODAL 4 /* linear forms only: P1-interpolated coefficient */
#define FS_COEF_CELL_TENSOR 6 /* stiffness only: data[n_cells][9], one row-major 3x3 tensor per cell (2-D: leading 2x2 block) -
                               * an Expression of degree 0 with a tensor value, examples/test_heat_transfer.py:90 */
#define FS_COEF_CELL_QP 7 /* stiffness on CG2 spaces only: data[n_cells][14], the coefficient at the 14 points of the degree-5 rule on
                           * tetrahedra / at the 6 points of the degree-4 rule on triangles (entries 0..5; the rest unused): a
                           * conductivity that depends on the P2 temperature iterate, evaluated where the integrand is */
#define FS_COEF_CELL_ROW 5 /* advection velocity only: data[n_cells][d+1][3], V_a = (d+1)/|K| int_K u phi_a dx per cell and test
                            * function - integrates inner(u, grad T) q dx exactly for a finite-element velocity u */

typedef struct fs_coef {
    int mode;             /* FS_COEF_* */
    double value;         /* FS_COEF_CONST */
    const double* data;   /* FS_COEF_CELL: [n_cells]; FS_COEF_NODAL: [n_dofs_local] (host) */
    double tensor[9];     /* FS_COEF_TENSOR, row-major */
} fs_coef;

/* Bilinear form  a(u,v) = int stiffness * grad u . grad v dx + int mass * u v dx
 * (scalar space: ScalarTransportSolver.py:284-285 and the 1/dt capacity term
 * :292), or for a vector space the isotropic elasticity operator
 * int (2 mu sym grad u + lambda div u I) : grad v dx  (LinearElasticitySolver.py:62-69, 215)
 * plus  mass * u . v. */
typedef struct fs_bilinear_form {
    fs_coef stiffness;   /* scalar spaces */
    fs_coef mass;        /* both */
    double lame_mu;      /* vector spaces */
    double lame_lambda;  /* vector spaces */
    /* scalar spaces: + advection_scale * int (v . grad u) q dx  (ScalarTransportSolver.py:311; Galerkin,
     * non-symmetric).  v: FS_COEF_CONST -> tensor[0..2]; FS_COEF_CELL -> data[n_cells][3]; FS_COEF_CELL_ROW -> data[n_cells][d+1][3]. */
    fs_coef advection;
    double advection_scale;
    /* SUPG ("SPUG" in the reference, ScalarTransportSolver.py:259-270): the test function becomes
     * q + tau (v . grad q), tau = 0.5 h / (4/(Pe h) + 2|v|), h = 2 * circumradius of the cell.  supg_pe > 0 adds the
     * tau-part to the advection term (streamline diffusion) and to the mass term; v is the advection velocity
     * (set advection_scale = 0 to get the mass part only, e.g. for the old-step operator).  CG1 and CG2 scalar spaces; on CG2 the
     * diffusion term changes too: grad(q + tau v . grad q) = grad q + tau H_q v with the cell-wise constant Hessian of q. */
    double supg_pe;
} fs_bilinear_form;

/* Replaces dolfin.assemble(a) / the matrix half of assemble_system: numeric
 * cell loop (tabulate_tensor + MatSetValues(ADD)).  add == 0 zeroes A first.  The host arrays of the form have been consumed
 * when the call returns; the cell loop itself is enqueued on the library's stream and the call does NOT wait for it (what the
 * caller does next - the load vector, the Dirichlet rows - is prepared meanwhile; every call that hands data back waits). */
int fs_assemble_matrix(fs_matrix_t A, const fs_bilinear_form* form, int add);

/* Linear form  L(v) = int source * v dx  (body source, ScalarTransportSolver.py:213-226;
 * vector spaces: constant body force f, LinearElasticitySolver.py:227-228, in
 * vector_value) + int div_coef * div v dx (vector spaces only: the thermal-stress load
 * E alpha (T-T0)/(1-2nu) I : grad v, LinearElasticitySolver.py:78-85, 231-238; a nodal
 * coefficient enters by its cell mean, which is what one-point quadrature of a P1 field gives).
 * add == 0 zeroes b first. */
typedef struct fs_linear_form {
    fs_coef source;
    double vector_value[3];
    fs_coef div_coef;
    fs_coef supg_velocity;   /* with supg_pe > 0: + int source * tau (v . grad q) dx (constant, per-cell and nodal sources;
                              * constant / per-cell velocity; nodal: exact for the source's P1 / P2 interpolant) */
    double supg_pe;
} fs_linear_form;
int fs_assemble_vector(fs_space_t space, const fs_linear_form* form, fs_vector_t b, int add);

/* Right-hand side of the L2 projection of the von Mises stress onto the scalar CG1 space of the mesh
 * (LinearElasticitySolver.py:71-76, project(sqrt(3/2 s:s), FunctionSpace(mesh, 'P', 1))): b_a = int vm(u) phi_a dx with
 * s = dev(2 mu sym(grad u) + lambda div u I).  disp_space: 3-vector CG1 or CG2 space, u its dof vector (owned + ghost);
 * p1_space: scalar CG1 space on the SAME mesh.  The projection itself is fs_assemble_matrix(mass = 1) on p1_space +
 * fs_krylov_solve.  CG1 displacement: exact; CG2: 4-point degree-2 rule. */
int fs_assemble_von_mises(fs_space_t disp_space, fs_vector_t u, double mu, double lambda, fs_space_t p1_space, fs_vector_t b);

/* Right-hand sides of the L2 projection of the fluid stress  nu (grad u + grad u^T) - p I  onto CG1
 * (CoupledNavierStokesSolver.py:149-155, viscous_stress): b[vertex*9 + 3 i + j] = int sigma_ij phi_vertex dx for a
 * Taylor-Hood iterate w (block (u_x,u_y,u_z,p) per CG2 node).  Each of the 9 components is then one CG1 mass-matrix
 * solve (fs_assemble_matrix(mass = 1) on p1_space + fs_krylov_solve).  On triangles (2-D Taylor-Hood, third slot of the
 * block unused): b[vertex*4 + 2 i + j], four components. */
int fs_assemble_viscous_stress(fs_space_t th_space, fs_vector_t w, double nu, fs_space_t p1_space, fs_vector_t b);
/* The same with nu(p) = nu (p / p_ref)^exponent (fs_ns_form.viscosity_pressure_ref / _exponent). */
int fs_assemble_viscous_stress_nn(fs_space_t th_space, fs_vector_t w, double nu, fs_space_t p1_space, fs_vector_t b,
                                  double viscosity_pressure_ref, double viscosity_pressure_exponent);

/* SUPG part of the boundary integrals (the reference substitutes q + tau (v . grad q) in them too,
 * ScalarTransportSolver.py:296-298 with Tq): for every listed boundary facet (cell behind it, local vertex opposite)
 *   b_a += g_f * area * w_a                      (flux / Neumann / HTC ambient loads; g may be NULL)
 *   A_ab += h_f * (area / 3) * w_a, b on the facet (HTC / Robin matrices; h may be NULL)
 * for all four vertices a of the cell, w_a = tau (v . grad phi_a).  A or b may be NULL.  CG2 spaces: a runs over all dofs of
 * the cell, the load takes v . grad q_a at the facet centroid, the matrix  h tau int_F phi_b (v . grad q_a) ds  by quadrature. */
int fs_assemble_facet_supg(fs_space_t space, fs_matrix_t A, fs_vector_t b, int64_t n_facets, const int32_t* facet_cell,
                           const int32_t* facet_opposite, const double* g, const double* h, const fs_coef* velocity,
                           double supg_pe);

/* Boundary-facet integrals over an explicit facet list (the host resolves
 * ds(id) to facets).  tri[n_facets][3] local vertex ids.
 *   vector:  b_a += int g phi_a ds, g[n_facets][ncomp]  (flux / Neumann / traction,
 *            ScalarTransportSolver.py:176-200, LinearElasticitySolver.py:165-196)
 *   matrix:  A_ab += int h phi_a phi_b ds, h[n_facets]   (HTC / Robin, :201-208) */
int fs_assemble_facet_vector(fs_space_t space, int64_t n_facets, const int32_t* tri,
                             const double* g, fs_vector_t b);
int fs_assemble_facet_matrix(fs_matrix_t A, int64_t n_facets, const int32_t* tri, const double* h);

/* Replaces DirichletBC.apply(A, b) (symmetric == 0: rows -> identity, b_i = g;
 * LinearVariationalSolver, SolverBase.py:608-612) and the elimination done by
 * assemble_system (symmetric == 1: additionally b -= A[:,i] g, columns zeroed;
 * SolverBase.py:644).  dofs are local dof indices (may include ghosts); later
 * entries win on duplicates, as later BCs do in DOLFIN.  A may be NULL to set
 * only b_i = g. */
int fs_apply_dirichlet(fs_matrix_t A, fs_vector_t b, int64_t n, const int32_t* dofs,
                       const double* vals, int symmetric);

/* y = A x  (PETSc MatMult).  x has n_cols entries, y n_rows.  With a halo plan
 * attached and a communicator up, ghosts of x are refreshed first. */
int fs_spmv(fs_matrix_t A, fs_vector_t x, fs_vector_t y);

/* Matrix-free product y = K(form) x on a scalar CG1 or (round 6: constant / per-cell scalar coefficients, no advection) CG2 space over tetrahedra (what `assemble(a)` followed by MatMult
 * would give, SolverBase.py:586-590, without forming the matrix): the row-gather assembly walk with every local row
 * multiplied into x.  No Dirichlet rows (callers mask); x must carry current ghost values.  reps > 1 with
 * ms_per_launch != NULL additionally times reps launches with HIP events (mean milliseconds per product). */
int fs_operator_apply(fs_space_t V, const fs_bilinear_form* form, fs_vector_t x, fs_vector_t y, int reps,
                      double* ms_per_launch);

/* ---- Krylov (PETScKrylovSolver("cg", pc).solve, SolverBase.py:663-670) -------- */

#define FS_KSP_CG 0
#define FS_KSP_BICGSTAB 1 /* non-symmetric operators (advection); PETSc KSPBCGS, right Jacobi */
#define FS_PC_NONE 0
#define FS_PC_JACOBI 1
#define FS_NORM_UNPRECONDITIONED 0 /* ||b - A x||_2 <= rtol ||b||_2 (BASELINE.json's definition) */
#define FS_NORM_PRECONDITIONED 1   /* ||D^-1 (b - A x)||_2 <= rtol ||D^-1 b||_2: PETSc's KSPCG default, robust when
                                    * constrained (identity) rows and physical rows differ by orders of magnitude */

typedef struct fs_krylov_opts {
    int method;          /* FS_KSP_CG | FS_KSP_BICGSTAB */
    int precond;         /* FS_PC_* */
    double rtol;         /* stop when ||r||_2 <= max(rtol*||b||_2, atol) */
    double atol;
    int max_iter;
    int batch;           /* iterations enqueued between host polls (0 = default 32) */
    int nonzero_guess;   /* 0: x0 = 0 (PETSc default); 1: use x on entry */
    int norm_type;       /* FS_NORM_*; the preconditioned norm needs CG + Jacobi + diagonal_scale */
    int diagonal_scale;  /* CG + Jacobi only: run on D^-1/2 A D^-1/2 (PETSc KSPSetDiagonalScale): same iterates,
                          * 25 % less vector traffic; A itself is left untouched (a scaled copy is kept) */
    int pipelined;       /* CG + Jacobi + diagonal_scale only.  1: the pipelined recurrence of Ghysels & Vanroose - the sums
                          * of an iteration are all-reduced WHILE its product runs instead of between product and update
                          * (same iterates in exact arithmetic, 112 instead of 72 B/DOF of vector traffic, attainable
                          * accuracy guarded by the same true-residual restarts); 0 / -1: the single-reduction recurrence
                          * (the default: measured on MI355X the pipelined one is the slower of the two at every size and
                          * transport tried, DESIGN.md section 5 - it is kept for communicators with a slow all-reduce) */
} fs_krylov_opts;

typedef struct fs_krylov_stats {
    int iterations;
    int converged;          /* 1 converged, 0 max_iter reached, -1 breakdown */
    double bnorm;           /* ||b||_2 (global) */
    double rel_residual;    /* recurrence ||r||/||b|| at exit */
    double true_rel_residual; /* ||b - A x||/||b|| recomputed at exit */
    double solve_ms;        /* wall time of the solve (host clock, synchronised) */
    double spmv_ms;         /* mean duration of the fused SpMV+dots kernel (HIP events) */
    double update_ms;       /* mean duration of the fused vector-update kernel */
    int64_t spmv_bytes;     /* algorithmic bytes of one SpMV: nnz*12 + n*20 */
    int row_classes;        /* > 0: the product ran in row-dictionary form with this many distinct rows (the operator of a uniform
                             * box mesh with constant coefficients: class numbers + the distinct rows in LDS instead of the value
                             * stream, verified bit for bit against the assembled values); 0: the streaming kernels */
    int fused_iteration;    /* 1: every CG iteration was ONE launch (update of iteration k + product of iteration k + 1 on a
                             * row-dictionary operator; spmv_ms is then the duration of that launch and update_ms 0); 2: a decomposed
                             * space - that launch preceded by the peer-to-peer exchange kernel (two launches per iteration;
                             * spmv_ms is the pair) */
    int classes_kept;       /* 1: the class table of the previous call on this space was kept - every row of THIS matrix was
                             * compared with its old class, bit for bit, and none differed (a steady problem solved again, a
                             * transient one with a constant step: one pass over the values instead of three); 0: the classes
                             * were found from scratch (or the streaming kernels ran) */
    int lattice_order;      /* 1: a scalar CG2 operator on a uniform box, solved in the solver's lattice order of the half grid (values,
                             * b and x permuted in and out; the API numbering is untouched): option "lattice_order" */
    int launches;           /* iterations ENQUEUED over all passes (fs_krylov_solve): those behind the one that stopped the
                             * recurrence return on the status word - launches - iterations of them, a few microseconds each */
    int product_kind;       /* kernel family of the solve's products (fs_last_product_kind): 0 streaming, 1 row-dictionary work items
                             * (also inside the one-launch iteration), 2 lattice tiles, 3 marching windows of a P1 box, 4 block rows,
                             * 5 marching windows of a CG2 box in lattice order */
} fs_krylov_stats;

int fs_krylov_solve(fs_matrix_t A, fs_vector_t b, fs_vector_t x, const fs_krylov_opts* opts,
                    fs_krylov_stats* stats);
/* ||r_k||_2^2 history of the last solve, k = 0..iterations. */
int fs_krylov_history(double* out, int capacity, int* count);

/* Time |reps| back-to-back launches of the SpMV kernel with HIP events on the
 * library's stream; returns mean milliseconds per launch.  reps > 0: bare y = A x;
 * reps < 0: the CG flavour fused with the three dot products (y plays r). */
int fs_spmv_benchmark(fs_matrix_t A, fs_vector_t x, fs_vector_t y, int reps, double* ms_per_launch);

/* y = A x (MatMult, behind KSPSolve at SolverBase.py:663-670) through the ROW-DICTIONARY form of the product where the rows of A
 * repeat (a uniform box mesh with constant coefficients): the classes are found from A's current values inside this call (or
 * kept from the previous call on the same space, where every row of A still equals its old class),
 * every row is verified against its class, and no later call multiplies from the table without verifying its own matrix against it.  *row_classes = number of distinct
 * rows used, 0 = the rows do not repeat and the streaming product ran.  Both forms give the same bits as fs_spmv. */
int fs_spmv_dictionary(fs_matrix_t A, fs_vector_t x, fs_vector_t y, int* row_classes);

/* Which kernel the LAST product launched by this process went through (MatMult, SolverBase.py:663-670; a diagnostic - PETSc's
 * analogue is the -log_view line of MatMult): 0 streaming SELL / DIA kernels, 1 row-dictionary work items (k_dict_spmv), 2 lattice
 * tiles of a CG2 box (k_lattice_spmv), 3 marching windows of a P1 box (k_box_spmv, round 6: options "box_spmv" 1 / 0 and
 * "box_min_rows", default 1 500 000), 4 block-row dictionary (k_dict_spmv3), 5 marching windows of a CG2 box in lattice order
 * (k_lat_march, round 6: option "lattice_march" 1 / 0).  The one-launch iteration k_dict_cg_iter does not count as a product here. */
int fs_last_product_kind(void);

/* ---- smoothed-aggregation AMG (PETScPreconditioner("petsc_amg") + set_near_nullspace,
 *      SolverBase.py:643-672; Chebyshev/Jacobi level smoother as the PETScOptions there ask) ---- */

typedef struct fs_amg_s* fs_amg_t;

typedef struct fs_amg_opts {
    double strength_threshold; /* theta of |A_ij|^2 > theta^2 |A_ii||A_jj| (block Frobenius norms); 0 = 0.05,
                                * negative = every coupling above rounding noise (1e-8) */
    int max_levels;            /* 0 = 10 */
    int coarse_size;           /* stop coarsening at this many dofs; 0 = 500 */
    int smoother_steps;        /* Chebyshev steps per pre/post smoothing; 0 = 2 (PETSc mg_levels_ksp_max_it) */
    int eig_steps;             /* power-iteration steps before the Rayleigh quotient of the eigenvalue estimate; 0 = 15 */
    int rigid_body_modes;      /* nullspace == NULL and a 3-vector CG1 space: build the six rigid-body modes of
                                * build_nullspace() (SolverBase.py:674-706) on the device from the node coordinates */
} fs_amg_opts;

/* Build the hierarchy for the assembled (Dirichlet-eliminated, SPD) matrix.  nullspace: host array
 * [n_nullspace][n_dofs] of near-null-space vectors (the rigid-body modes of build_nullspace(),
 * SolverBase.py:674-706), or NULL = one constant per component.  The matrix must outlive the
 * hierarchy and keep its values.  On a space with ghost nodes the hierarchy is built on this rank's diagonal
 * block (ghost columns dropped): fs_amg_solve is then CG on the distributed operator with the rank-local
 * V-cycles as non-overlapping additive Schwarz preconditioner. */
int fs_amg_setup(fs_matrix_t A, int n_nullspace, const double* nullspace, const fs_amg_opts* opts, fs_amg_t* out);
/* Distributed fine level under a replicated hierarchy (several GPUs; PETSc's GAMG under mpirun keeps the fine level distributed
 * and agglomerates the coarse ones, SolverBase.py:643-672 + :634).  M was set up on the UNDECOMPOSED operator, identically on
 * every rank; A_local holds this rank's rows of the decomposed operator (halo plan on its space); owned_global_nodes[i] = the
 * node, in the undecomposed space, of local owned node i.  Afterwards fs_amg_apply / fs_amg_solve smooth, restrict and prolong
 * level 0 on this rank's rows (ghost refresh before every fine product, the restricted right-hand side summed over the ranks)
 * and apply levels >= 1 as they are: the preconditioner - and the iteration count - of one GPU, with the fine-level work
 * divided by the number of ranks.  COLLECTIVE in its use (every rank attaches its part before the first solve).  A second call
 * on a hierarchy that has its fine level attached swaps in A_local - a matrix on the SAME decomposed space holding the same
 * operator (a hierarchy kept over several solves multiplies with the caller's current matrix, not the first one). */
int fs_amg_attach_distributed_fine(fs_amg_t M, fs_matrix_t A_local, int64_t n_owned_nodes, const int32_t* owned_global_nodes);
int fs_amg_destroy(fs_amg_t amg);
int fs_amg_info(fs_amg_t amg, int* n_levels, double* operator_complexity, double* grid_complexity, double* setup_ms);
int fs_amg_level_info(fs_amg_t amg, int level, int64_t* n_nodes, int* block_size, int64_t* nnz_blocks,
                      int64_t* p_nnz_blocks, int* p_block_cols, double* lambda_max);
/* which: 0 = level operator A (block CSR, blocks row-major), 1 = prolongator to this level from the
 * next, 2 = near-null space [n_dofs][nb] (val only).  Test/inspection hook. */
int fs_amg_level_get(fs_amg_t amg, int level, int which, int32_t* rowptr, int32_t* col, double* val);
/* z = M r: one V-cycle from a zero guess (PCApply). */
int fs_amg_apply(fs_amg_t amg, fs_vector_t r, fs_vector_t z);
/* CG preconditioned by the V-cycle (PETScKrylovSolver("cg", pc).solve, SolverBase.py:660-670);
 * uses rtol, atol, max_iter, nonzero_guess and norm_type of the options. */
int fs_amg_solve(fs_amg_t amg, fs_vector_t b, fs_vector_t x, const fs_krylov_opts* opts, fs_krylov_stats* stats);

/* ---- Taylor-Hood Navier-Stokes (CoupledNavierStokesSolver.py:288-381, 215-245, 492-528) -------------
 * Unknowns: one block (u_x, u_y, u_z, p) per CG2 node of fs_space_create(mesh, FS_FAMILY_CG, 2, 4); the
 * pressure is CG1, the pressure slot of an edge node is a dummy unknown with a unit row. */

typedef struct fs_ns_form {
    double kinematic_viscosity; /* nu: 2 nu eps(u):eps(v) */
    double density;             /* rho: -(p/rho) div v + (q/rho) div u */
    double inv_dt;              /* 1/dt of the backward-Euler term (F_transient), 0 = steady */
    double body_force[3];       /* f of -f.v (acceleration, "just gravity, without * rho") */
    int convection;             /* (grad(u) u0).v with u0 = velocity part of w0 */
    int newton;                 /* add (grad(u0) u).v to J and (grad(u0) u0).v to g: derivative(action(F, w0)) */
    double mesh_velocity[3];    /* ALE frame (reference_frame_settings, CoupledNavierStokesSolver.py:321-329): the advecting
                                 * velocity is u0 - mesh_velocity, i.e. the term is (grad(u) (u0 - w)).v; constant vector.
                                 * Written for the new iterate, the Newton right-hand side keeps (grad(u0) u0).v. */
    int g2_mode;                /* G2 streamline term of advection_settings (CoupledNavierStokesSolver.py:334-363):
                                 * F -= delta1 (a.grad u).(a.grad v) dx, a the advecting velocity, h = 2 circumradius;
                                 * 0 off, 1: delta1 = kappa1 h^2 (Re <= 1), 2: kappa1/2 h/|a| (steady) or
                                 * kappa1/2 / sqrt(1/dt^2 + 1/(|a|^2 h^2)) (inv_dt > 0).  Enters J with a frozen at w0 (the
                                 * system is written for the new iterate, so g is unchanged and J w - g is the exact residual). */
    double g2_kappa1;
    double viscosity_pressure_ref;      /* > 0: the non-Newtonian law of CoupledNavierStokesSolver.viscosity (:194-213, the branch */
    double viscosity_pressure_exponent; /* without a temperature): nu(p) = nu (p / p_ref)^exponent with the pressure of w0 (needs
                                         * w0; Picard in the viscosity - J w - g stays the exact residual).  0: Newtonian. */
} fs_ns_form;

/* The non-Newtonian laws of CoupledNavierStokesSolver.viscosity (CoupledNavierStokesSolver.py:194-213), attached to the
 * Taylor-Hood space and used by every routine that evaluates nu on it (fs_assemble_navier_stokes, the pressure-boundary traction
 * term, fs_assemble_viscous_stress*), in place of the (p_ref, exponent) pair those calls carry:
 *   kind 1  nu (p / pressure_ref)^pressure_exponent                                          (:205-207, no temperature)
 *   kind 2  nu (1 + pressure_coef p / pressure_ref) (1 - temperature_coef T / temperature_ref)   (:199-203, solving_temperature)
 * p: the pressure of the state w0 the call linearises at; temperature: CG1 field stored like the pressure inside w0 - one value per
 * LOCAL NODE of the Taylor-Hood space, read at the vertex nodes only (one GPU: the vertices are the first nodes; a decomposed space
 * orders owned vertices, owned edges, ghost vertices, ghost edges).  The vector is read at assembly time - keep it alive and current;
 * nothing is copied.  law = NULL or kind 0 detaches. */
typedef struct fs_viscosity_law {
    int kind;
    double pressure_ref, pressure_exponent;
    double pressure_coef, temperature_coef, temperature_ref;
    fs_vector_t temperature;
} fs_viscosity_law;
int fs_space_set_viscosity_law(fs_space_t th_space, const fs_viscosity_law* law);

/* J <- linearised operator at w0, g <- right-hand side such that J w_new = g is the Newton (or Picard) step
 * written for the new iterate.  w_prev: previous time step (may be NULL when inv_dt = 0). */
int fs_assemble_navier_stokes(fs_matrix_t J, fs_vector_t g, fs_vector_t w0, fs_vector_t w_prev, const fs_ns_form* form);

/* Pressure boundaries, added to J and g after fs_assemble_navier_stokes and before the Dirichlet rows:
 *   F += inner(p_b n, v) ds - nu inner((grad(u) + grad(u)^T) n, v) ds   (CoupledNavierStokesSolver.py:449-453)
 * facet_cell / facet_opposite: the cell behind each boundary facet and the local vertex opposite to it;
 * facet_value: p_b per facet, NULL = the pressure 'farfield' type (traction term only, :459-460). */
int fs_assemble_ns_pressure_boundary(fs_matrix_t J, fs_vector_t g, int64_t n_facets, const int32_t* facet_cell,
                                     const int32_t* facet_opposite, const double* facet_value, double kinematic_viscosity);
/* The same with the pressure-dependent viscosity of fs_ns_form (w0: the state whose pressure enters nu; ref 0 = off) and,
 * with values_per_facet = 3, a boundary pressure that varies over the facet: facet_value[f][k] = p_b at the facet's k-th
 * vertex in the order of the cell's local vertices (the opposite one left out) - the P1 interpolant DOLFIN evaluates a
 * degree-1 Expression with; values_per_facet = 1: one value per facet. */
int fs_assemble_ns_pressure_boundary_nn(fs_matrix_t J, fs_vector_t g, int64_t n_facets, const int32_t* facet_cell,
                                        const int32_t* facet_opposite, const double* facet_value, double kinematic_viscosity,
                                        fs_vector_t w0, double viscosity_pressure_ref, double viscosity_pressure_exponent,
                                        int values_per_facet);

typedef struct fs_saddle_opts {
    double rtol, atol;          /* on ||g - J w||_2 (relative to ||g||_2) */
    int max_iter;               /* 0 = 600 */
    int restart;                /* FGMRES restart length, 0 = 60 */
    double kinematic_viscosity, density, inv_dt; /* of the Schur-complement approximation */
    int velocity_sweeps;        /* Jacobi sweeps standing in for A^-1, 0 = 1 */
    double inner_rtol;          /* of the pressure Laplacian / mass CG solves, 0 = 1e-2 */
    int nonzero_guess;
} fs_saddle_opts;

/* Restarted FGMRES on the coupled system (the reference lets PETSc LU do this, SolverBase.py:615-626),
 * right-preconditioned by [A 0; D S]^-1 with the Cahouet-Chabard Schur complement
 * S^-1 = rho^2 ((1/dt) Kp^-1 + nu Mp^-1).  Kp: CG1 stiffness matrix (coefficient 1) with the pressure
 * Dirichlet dofs eliminated, may be NULL when inv_dt = 0; Kp_amg: optional hierarchy of Kp (fs_amg_setup) - one
 * V-cycle then stands in for Kp^-1 instead of an inner CG solve; Mp: CG1 mass matrix.
 * Several GPUs (J, Kp, Mp on decomposed spaces): the multi-dot sums are all-reduced, the halo of the preconditioned
 * vector is exchanged twice per iteration, M_p^-1 acts on the rank-local block; Kp_amg is either the hierarchy of the
 * GLOBAL pressure Laplacian held by every rank (its row count differs from the owned vertices: the pressure
 * residual is all-reduced into the global vector and the V-cycle replicated - same operator as on one GPU) or a
 * rank-local hierarchy (then an inner additive-Schwarz CG solve to inner_rtol). */
int fs_saddle_solve(fs_matrix_t J, fs_matrix_t Kp, fs_amg_t Kp_amg, fs_matrix_t Mp, fs_vector_t b, fs_vector_t x,
                    const fs_saddle_opts* opts, fs_krylov_stats* stats);

/* ---- multi-GPU (MPI inside PETSc/DOLFIN under mpirun; SolverBase.py:102-118, 634) */

#define FS_UNIQUE_ID_BYTES 128
int fs_comm_get_unique_id(char id[FS_UNIQUE_ID_BYTES]); /* rank 0, then broadcast out of band */
int fs_comm_init(int n_ranks, int rank, const char id[FS_UNIQUE_ID_BYTES]);
int fs_comm_info(int* n_ranks, int* rank);
int fs_comm_allreduce_sum(double* host_inout, int n); /* utility: host scalars */
/* All-gather of host arrays: every rank contributes n_send <= n_max doubles; recv [n_ranks][n_max] (padding undefined).
 * One ncclAllGather; the solver API gathers the owned parts of a solution with it. */
int fs_comm_allgather(const double* host_send, int64_t n_send, int64_t n_max, double* host_recv);
int fs_comm_finalize(void);

/* Halo plan of a space (PETSc VecScatter ghost update): for each neighbour the
 * owned local dofs to send (concatenated in send_idx) and the number of ghosts
 * received; ghosts are stored after the owned dofs, neighbour by neighbour in
 * the order given. */
int fs_space_set_halo(fs_space_t space, int n_neighbors, const int32_t* neighbor_ranks,
                      const int64_t* send_counts, const int32_t* send_idx,
                      const int64_t* recv_counts);
/* Refresh the ghost entries of a local vector (n_dofs_local long). */
/* The same with an explicit scatter list: the k-th value received from a neighbour goes to local dof recv_idx[k]
 * (>= the owned dofs).  For layouts whose ghosts are not grouped by owner, e.g. CG2 nodes
 * [owned vertices | owned edges | ghost vertices | ghost edges]. */
int fs_space_set_halo_indexed(fs_space_t space, int n_neighbors, const int32_t* neighbor_ranks, const int64_t* send_counts,
                              const int32_t* send_idx, const int64_t* recv_counts, const int32_t* recv_idx);
int fs_halo_exchange(fs_space_t space, fs_vector_t v);
/* Opt-in ghost refresh WITHOUT the library in the data path, for the ranks of one node: every rank owns a fine-grained
 * receive buffer and arrival flags, its neighbours map them through hipIpc, a kernel gathers the values a neighbour
 * needs and stores them straight into that neighbour's buffer (over xGMI), then releases a sequence number; the
 * receiver's kernel waits for the sequence numbers of all its neighbours and fills the ghost entries.  Same place in
 * the iteration as the grouped ncclSend / ncclRecv it replaces (PETSc's VecScatter behind MatMult,
 * SolverBase.py:634 under mpirun), about a third of its latency on MI355X.
 * The call ends with a self-test of the mapped buffers (an all-reduce and a ghost refresh of known values, 0.5 s time-out)
 * whose outcome the ranks agree on: FS_ERR_COMM on every rank if any of them saw a wrong or missing value.
 * COLLECTIVE over the communicator: every rank calls it for its space in the same order; enable = 0 turns it off
 * (required before fs_space_set_halo replaces the plan).  FS_ERR_COMM without a communicator.
 * Destroying a space with the exchange still on frees buffers its neighbours have mapped: do it on every rank at the same
 * point of the program (after a solve returned no rank has a store in flight), or turn the exchange off first. */
int fs_space_enable_p2p_halo(fs_space_t space, int enable);
/* Mean latency (ms) of the two collectives of a distributed CG iteration as the solver issues them - the 3-double
 * ncclAllReduce behind VecDot (SolverBase.py:634 under mpirun) and the ghost refresh of this space's halo plan behind
 * MatMult's VecScatter - over `reps` back-to-back calls (HIP events).  Collective; zeros on one rank. */
int fs_comm_benchmark(fs_space_t space, int reps, double* allreduce_ms, double* halo_ms);

#ifdef __cplusplus
}
#endif
#endif /* FENICSSOLVER_AMD_H */
