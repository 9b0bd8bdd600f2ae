#include <stdio.h>
#include <stdlib.h>
#include "p264pipe.h"
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *b = malloc(n); fread(b, 1, n, f); fclose(f);
    int reps = argc > 2 ? atoi(argv[2]) : 5;
    double best = 0;
    for (int r = 0; r < reps; r++) {
        p264pipe *p = p264pipe_open(-1, 1, 1);
        p264pipe_set_input(p, 0, b, n);
        p264pipe_stats_t st; p264pipe_run(p, 0, &st);
        if (st.pictures / st.seconds > best) best = st.pictures / st.seconds;
        p264pipe_close(p);
    }
    printf("%.1f fps\n", best);
    return 0;
}
