/*
 * ref_mpi_dump.c -- TEST INFRASTRUCTURE (oracle/): a dump harness around the reference's
 * attention-mpi.c.  The reference file is compiled UNMODIFIED from where it lies under
 * /root/reference (it is #included by path, nothing is copied into this repository); only its
 * main() is renamed so that this main() can call the reference's own
 *     attention(Q, K, V, result, m, n, dk, dv, mpi_rank, mpi_size)      attention-mpi.c:191-407
 * and write the raw `result` array -- the output of the reference's fp32 K/V-sharded pipeline,
 * before verify()'s +-0.02 hides everything -- to a file.
 *
 *     mpiexec -n P oracle/_ref/attention-mpi-dump <case.bin> <out.f32>
 *
 * Output: m*dv float32 (the reference widens fp32 values to fp64 at :373/:396, so every result
 * value is exactly representable in fp32; stored narrow to keep the fixtures small).
 * Built by `make -C oracle ref`; used by oracle/make_ref_mpi_fp32.py to produce
 * tests/golden/ref_mpi_fp32/.  Nothing in the product references this file.
 */
#define main ref_mpi_main
#include REF_SOURCE
#undef main

int main(int argc, char *argv[])
{
    if (argc < 3) {
        fprintf(stderr, "Usage: %s <testing data> <out.f32>\n", argv[0]);
        return 1;
    }
    int rank, size;
    MPI_Init(&argc, &argv);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    double *Q = NULL, *K = NULL, *V = NULL, *result = NULL;
    int m, n, dk, dv;
    if (rank == 0) {                       /* as the reference's main(): attention-mpi.c:513-517 */
        read_matrices(argv[1], &Q, &K, &V, &m, &n, &dk, &dv);
        result = malloc(sizeof(double) * m * dv);
    }
    attention(Q, K, V, result, m, n, dk, dv, rank, size);
    int rc = 0;
    if (rank == 0) {
        FILE *f = fopen(argv[2], "wb");
        if (!f) {
            fprintf(stderr, "cannot write %s\n", argv[2]);
            rc = 1;
        } else {
            for (long i = 0; i < (long)m * dv; ++i) {
                float x = (float)result[i];
                if ((double)x != result[i] && !(result[i] != result[i])) {
                    fprintf(stderr, "result[%ld] is not an fp32 value\n", i);
                    rc = 1;
                }
                fwrite(&x, sizeof x, 1, f);
            }
            fclose(f);
        }
    }
    MPI_Finalize();
    free(Q); free(K); free(V); free(result);
    return rc;
}
