/* power.h -- board power sampling while a solve runs.
 *
 * The reference forks a shell loop that appends `nvidia-smi --query-gpu=power.draw` to a file once a second
 * (cpower.h:87-133 builds the command, solvers.c:537-546 brackets the solve with c_popen / c_pclose; flags
 * -p <file> and -id <device>, solvers.c:118-125).  Here a thread of the driver reads the amdgpu hwmon node
 * of the GPU (power1_average, or power1_input on parts that only report the instantaneous value; microwatts)
 * and appends one number in watts per line to the same kind of file; no child process, 10 samples a second. */
#ifndef MK_HOST_POWER_H
#define MK_HOST_POWER_H

typedef struct power_sampler power_sampler;

/* pci_bus_id as hipDeviceGetPCIBusId prints it ("0000:c1:00.0"); returns 0, or non-zero with a message on
 * stderr when the node cannot be found or read (the caller goes on without sampling) */
int power_sampler_start(const char *pci_bus_id, const char *out_path, int period_ms, power_sampler **out);
/* stops the thread, closes the file; any of the outputs may be NULL */
int power_sampler_stop(power_sampler *s, double *mean_watts, double *energy_joules, int *nsamples);

#endif
